#!/bin/bash
# Build an EXPERIMENTAL copy of the engine from the tree + one patch of scripts/exp/patches/, without touching the tree:
#   scripts/exp/build_variant.sh part_c_stamps   ->  limitador_amd/lib/variants/librl_engine_part_c_stamps.so
# An experiment script on the GPU box (a throw-away snapshot) copies the variant over limitador_amd/lib/librl_engine.so
# before it runs bench.py; the product library in this tree is never replaced.  Variants are not committed (lib/ is
# git-ignored) and should be deleted after the visit.
set -eu
name=$1; shift
root=$(cd "$(dirname "$0")/../.." && pwd)
tmp=$(mktemp -d)
mkdir -p "$tmp/limitador_amd" "$root/limitador_amd/lib/variants"
cp -r "$root/include" "$tmp/include"
cp -r "$root/limitador_amd/csrc" "$tmp/limitador_amd/csrc"
# (a variant that is only a set of -D flags has no patch file:  build_variant.sh partl_wpe4 -DRL_PARTL_WPE=4)
[ -f "$root/scripts/exp/patches/$name.patch" ] && (cd "$tmp" && patch -p1 < "$root/scripts/exp/patches/$name.patch")
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -DRL_EXPERIMENT "$@" -I"$tmp/include" "$tmp/limitador_amd/csrc/rl_engine.hip" \
    -o "$root/limitador_amd/lib/variants/librl_engine_$name.so"
rm -rf "$tmp"
ls -la "$root/limitador_amd/lib/variants/librl_engine_$name.so"
