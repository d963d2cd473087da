"""shim/gpu_storage.rs is source only (no rustc in this image): what CAN be checked without a compiler is that its
`extern "C"` block, its `#[repr(C)]` structs and the status constants it names still say what include/rl_engine.h says —
every function exists in the header with the same parameters (count, pointer-ness, constness, pointee), every struct has the
header's fields in the header's order, every constant the header's value."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = open(os.path.join(ROOT, "shim", "gpu_storage.rs")).read()
HDR = open(os.path.join(ROOT, "include", "rl_engine.h")).read()
HDR_NOCOMMENT = re.sub(r"/\*.*?\*/", " ", HDR, flags=re.S)

RUST_BASE = {"u8": "uint8_t", "u32": "uint32_t", "u64": "uint64_t", "i32": "int32_t", "c_char": "char", "c_int": "int",
             "std::ffi::c_void": "void", "c_void": "void", "RlEngine": "rl_engine", "RlConfig": "rl_config", "RlHit": "rl_hit",
             "RlLimitRow": "rl_limit_row", "RlCellRow": "rl_cell_row"}


def rust_type(t):
    """`*const RlHit` -> ('rl_hit', ['const']) : base type + one const/mut per pointer level, outermost first."""
    t = t.strip()
    quals = []
    while t.startswith("*"):
        m = re.match(r"\*(const|mut)\s+(.*)", t)
        quals.append(m.group(1))
        t = m.group(2).strip()
    return RUST_BASE[t], quals


def c_type(t):
    """`const rl_hit *hits` -> ('rl_hit', ['const']); `rl_engine **out` -> ('rl_engine', ['mut', 'mut'])."""
    t = re.sub(r"\b(struct|enum)\s+", "", t.strip())
    stars = t.count("*")
    t = t.replace("*", " ")
    words = t.split()
    const = "const" in words
    words = [w for w in words if w != "const"]
    base = words[0] if len(words) <= 2 else " ".join(words[:-1])  # the last word is the parameter's name (if any)
    if len(words) == 1:
        base = words[0]
    # (only the pointee of the innermost level is ever const in this header: `const T *p`)
    quals = ["mut"] * stars
    if const and stars:
        quals[-1] = "const"
    return base, quals[::-1] if False else quals


def shim_functions():
    block = re.search(r'extern "C" \{(.*?)\n\}', SHIM, flags=re.S).group(1)
    block = re.sub(r"#\[[^\]]*\]", " ", block)
    out = {}
    for m in re.finditer(r"fn\s+(\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", block, flags=re.S):
        params = [p.strip() for p in m.group(2).split(",") if p.strip()]
        out[m.group(1)] = ([rust_type(p.split(":", 1)[1]) for p in params], rust_type(m.group(3)) if m.group(3) else ("void", []))
    return out


def header_functions():
    out = {}
    for m in re.finditer(r"\n([A-Za-z_][\w \*]*?)\b(rl_\w+)\s*\(([^;{]*?)\)\s*;", HDR_NOCOMMENT, flags=re.S):
        params = [p.strip() for p in m.group(3).replace("\n", " ").split(",") if p.strip() and p.strip() != "void"]
        out[m.group(2)] = ([c_type(p) for p in params], c_type(m.group(1) + " x"))
    return out


def test_every_function_the_shim_declares_is_the_headers():
    shim, hdr = shim_functions(), header_functions()
    assert len(shim) >= 15, sorted(shim)
    for name, (params, ret) in shim.items():
        assert name in hdr, f"{name} is not declared in include/rl_engine.h"
        hp, hret = hdr[name]
        assert len(params) == len(hp), (name, len(params), len(hp))
        for i, (a, b) in enumerate(zip(params, hp)):
            assert a == b, f"{name}: parameter {i}: shim {a} vs header {b}"
        assert ret == hret, (name, ret, hret)


def test_repr_c_structs_have_the_headers_fields():
    for rs, c in (("RlConfig", "rl_config"), ("RlLimitRow", "rl_limit_row"), ("RlHit", "rl_hit"), ("RlCellRow", "rl_cell_row")):
        rbody = re.search(r"#\[repr\(C\)\][^{]*struct " + rs + r" \{(.*?)\}", SHIM, flags=re.S).group(1)
        rfields = [(n, RUST_BASE[t.strip()]) for n, t in re.findall(r"(\w+):\s*([\w:]+),", rbody)]
        cbody = re.search(r"typedef struct\s*(?:" + c + r")?\s*\{([^{}]*?)\}\s*" + c + ";", HDR_NOCOMMENT, flags=re.S).group(1)
        cfields = [(n, t) for t, n in re.findall(r"(\w+)\s+(\w+)\s*;", cbody)]
        assert rfields == cfields, (rs, rfields, cfields)


def test_constants_the_shim_names_have_the_headers_values():
    enum = dict((n, int(v)) for n, v in re.findall(r"\b(RL_[A-Z_]+)\s*=\s*(-?\d+)", HDR_NOCOMMENT))
    defines = dict((n, int(v.rstrip("uU"), 0)) for n, v in re.findall(r"#define\s+(RL_[A-Z_]+)\s+(0x[0-9A-Fa-f]+u?|\d+u?)", HDR))
    known = {**enum, **defines}
    seen = 0
    for name, val in re.findall(r"const\s+(RL_[A-Z_]+):\s*[iu]32\s*=\s*(-?[0-9xA-Fa-f_]+);", SHIM):
        assert name in known, name
        assert int(val.replace("_", ""), 0) == known[name], (name, val, known[name])
        seen += 1
    assert seen >= 6
