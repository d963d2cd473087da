"""In-tree build of the HIP engine (and, for the tests, of the CPU oracle).

``hipcc`` cross-compiles gfx950 without a GPU.  Outputs stay in-tree (``limitador_amd/lib``,
``oracle/``) so they travel with the repo snapshot to the GPU box; they are git-ignored.
"""
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "limitador_amd", "csrc")
# Two builds of the same sources, both in-tree:
#   limitador_amd/lib/       the release libraries (what bench.py, smoke() and a host link against): they read the
#                            documented handful of environment variables (include/rl_engine.h) and nothing else
#   limitador_amd/lib/exp/   the same libraries with -DRL_EXPERIMENT: every shape / budget / diagnostics switch is
#                            readable from the environment.  The test-suite loads these (tests/conftest.py sets
#                            LIMITADOR_AMD_LIB=exp) so that one pytest session can walk every engine mode;
#                            tests/test_gpu_release_lib.py runs the release build in a process of its own.
RELEASE_LIBDIR = os.path.join(ROOT, "limitador_amd", "lib")
EXP_LIBDIR = os.path.join(RELEASE_LIBDIR, "exp")
EXPERIMENT = os.environ.get("LIMITADOR_AMD_LIB", "") == "exp"
LIBDIR = EXP_LIBDIR if EXPERIMENT else RELEASE_LIBDIR
EXP_DEFS = ["-DRL_EXPERIMENT"] if EXPERIMENT else []
ENGINE_SO = os.path.join(LIBDIR, "librl_engine.so")
STORAGE_SO = os.path.join(LIBDIR, "librl_storage.so")
SHARDED_SO = os.path.join(LIBDIR, "librl_sharded.so")
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liblimitador_oracle.so")


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def _sources(dirpath, exts):
    out = []
    for base, _dirs, files in os.walk(dirpath):
        for f in files:
            if f.endswith(exts):
                out.append(os.path.join(base, f))
    return out


def build_engine(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> limitador_amd/lib/librl_engine.so"""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    srcs = _sources(CSRC, (".hip", ".hpp")) + [os.path.join(ROOT, "include", "rl_engine.h")]
    if not force and _newer(ENGINE_SO, srcs):
        return ENGINE_SO
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared"] + EXP_DEFS + [
           "-I" + os.path.join(ROOT, "include"), os.path.join(CSRC, "rl_engine.hip"), "-o", ENGINE_SO]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return ENGINE_SO


def build_storage(force=False, verbose=False):
    """g++ -> limitador_amd/lib/librl_storage.so (C++ host mirror of CounterStorage)."""
    src = os.path.join(CSRC, "host", "gpu_counter_storage.cpp")
    if not os.path.exists(src):
        return None
    srcs = _sources(os.path.join(CSRC, "host"), (".cpp", ".hpp", ".h")) + [os.path.join(ROOT, "include", "rl_engine.h"),
                                                                             os.path.join(ROOT, "include", "rl_storage.h"),
                                                                             os.path.join(ROOT, "include", "rl_ingest.h")]
    if not force and _newer(STORAGE_SO, srcs):
        return STORAGE_SO
    build_engine(force=False, verbose=verbose)
    ingest = os.path.join(CSRC, "host", "ingest.cpp")  # host-side ingest of the device matcher (rl_ingest.h)
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread"] + EXP_DEFS + ["-I" + os.path.join(ROOT, "include"), src,
           ingest, "-o", STORAGE_SO, "-L" + LIBDIR, "-lrl_engine", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return STORAGE_SO


def build_sharded(force=False, verbose=False):
    """g++ -> limitador_amd/lib/librl_sharded.so (include/rl_sharded.h: the routed multi-GPU step; links RCCL)."""
    src = os.path.join(CSRC, "host", "rl_sharded.cpp")
    srcs = [src, os.path.join(ROOT, "include", "rl_sharded.h"), os.path.join(ROOT, "include", "rl_engine.h")]
    if not force and _newer(SHARDED_SO, srcs) and _newer(SHARDED_SO, [ENGINE_SO] if os.path.exists(ENGINE_SO) else []):
        return SHARDED_SO
    build_engine(force=False, verbose=verbose)
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-D__HIP_PLATFORM_AMD__"] + EXP_DEFS + [
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(rocm, "include"), src, "-o", SHARDED_SO,
           "-L" + LIBDIR, "-lrl_engine", "-L" + os.path.join(rocm, "lib"), "-lamdhip64", "-ldl",
           "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(rocm, "lib")]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return SHARDED_SO


def build_oracle(force=False, verbose=False):
    """gcc -> oracle/liblimitador_oracle.so (test infrastructure, never used by the product)."""
    srcs = [os.path.join(ORACLE_DIR, "limitador_oracle.c"), os.path.join(ORACLE_DIR, "limitador_oracle.h")]
    if not force and _newer(ORACLE_SO, srcs):
        return ORACLE_SO
    cmd = ["make", "-C", ORACLE_DIR] + (["-B"] if force else [])
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, stdout=None if verbose else subprocess.DEVNULL)
    return ORACLE_SO


def build_exp(verbose=False):
    """The -DRL_EXPERIMENT libraries (limitador_amd/lib/exp/), in a process of its own: the build flavour is a
    module-level choice (LIMITADOR_AMD_LIB), so that a process only ever sees one set of libraries."""
    import sys

    env = dict(os.environ, LIMITADOR_AMD_LIB="exp")
    code = ("from limitador_amd import build as b; v=%r; b.build_engine(verbose=v); b.build_storage(verbose=v); "
            "b.build_sharded(verbose=v)" % bool(verbose))
    subprocess.run([sys.executable, "-c", code], check=True, cwd=ROOT, env=env)
    return EXP_LIBDIR


if __name__ == "__main__":
    print(build_engine(verbose=True))
    print(build_storage(verbose=True))
    print(build_sharded(verbose=True))
    print(build_oracle(verbose=True))
