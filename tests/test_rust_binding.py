"""bindings/rust/src/ffi.rs against include/rtiow_hip.h (there is no Rust toolchain in the image, so the
binding cannot be compiled here; this checks what a text comparison can: struct fields, layout constants,
function list)."""
import ctypes as C
import os
import re

from rtiow_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RS = open(os.path.join(ROOT, "bindings", "rust", "src", "ffi.rs")).read()
HDR = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "rtiow_hip.h")).read(), flags=re.S)

C_TO_RUST = {"double": "f64", "float": "f32", "int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "int64_t": "i64"}


def c_struct_fields(name):
    body = re.search(r"typedef struct \{([^}]*)\}\s*%s;" % name, HDR).group(1)
    out = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        ctype, rest = decl.split(None, 1)
        for item in rest.split(","):
            m = re.match(r"\s*(\w+)\s*(?:\[(\d+)\])?\s*$", item)
            out.append((m.group(1), C_TO_RUST[ctype] if not m.group(2) else f"[{C_TO_RUST[ctype]}; {m.group(2)}]"))
    return out


def rust_struct_fields(name):
    body = re.search(r"pub struct %s \{(.*?)\n\}" % name, RS, flags=re.S).group(1)
    return [(m.group(1), m.group(2).strip()) for m in re.finditer(r"pub (\w+):\s*([^,\n]+),", body)]


def test_structs_have_the_headers_fields_in_order():
    for name in ("rt_sphere", "rt_camera", "rt_params", "rt_stats"):
        assert rust_struct_fields(name) == c_struct_fields(name), name
        assert re.search(r"#\[repr\(C\)\]\s*(?:#\[derive\([^)]*\)\]\s*)?pub struct %s " % name, RS), f"{name} is not repr(C)"


def test_layout_constants_match_the_c_side():
    """The numbers asserted in ffi.rs are those of the ctypes mirrors tests/test_cabi.py checks."""
    sizes = {m.group(1): int(m.group(2)) for m in re.finditer(r"size_of::<(\w+)>\(\) == (\d+)", RS)}
    assert sizes == {"rt_sphere": C.sizeof(_ffi.rt_sphere), "rt_camera": C.sizeof(_ffi.rt_camera),
                     "rt_params": C.sizeof(_ffi.rt_params), "rt_stats": C.sizeof(_ffi.rt_stats)}
    assert (sizes["rt_sphere"], sizes["rt_camera"], sizes["rt_params"]) == (72, 152, 56)
    offs = {(m.group(1), m.group(2)): int(m.group(3)) for m in re.finditer(r"offset_of!\((\w+), (\w+)\) == (\d+)", RS)}
    for (st, field), off in offs.items():
        assert getattr(getattr(_ffi, st), field).offset == off, (st, field)
    assert offs[("rt_params", "t_min")] == 24 and offs[("rt_params", "seed")] == 32


def test_extern_block_declares_the_product_abi():
    product = re.sub(r"#ifdef RTIOW_CROSSCHECK_MODES.*?#endif", "", HDR, flags=re.S)
    want = sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", product)))
    got = sorted(re.findall(r"pub fn (rt_\w+)\(", RS))
    assert got == want
    assert re.search(r"RTIOW_HIP_ABI_VERSION: i32 = (\d+)", RS).group(1) == re.search(r"#define RTIOW_HIP_ABI_VERSION (\d+)", HDR).group(1)
    for k, v in re.findall(r"(RT_(?:OK|ERR_\w+)) = (-?\d+)", HDR):
        assert re.search(r"pub const %s: i32 = %s;" % (k, v), RS), k
    for k, v in re.findall(r"#define (RT_FLAG_\w+)\s+(0x[0-9a-f]+)u", HDR):
        assert re.search(r"pub const %s: u32 = %s;" % (k, v), RS), k
    # the two limits of the boundary
    assert re.search(r"#define RT_MAX_SPHERES \(1 << 24\)", HDR) and "pub const RT_MAX_SPHERES: i32 = 1 << 24;" in RS
    assert re.search(r"#define RT_SAMPLE_CLAMP 65536\.0", HDR) and "pub const RT_SAMPLE_CLAMP: f64 = 65536.0;" in RS


def test_shim_covers_the_reference_types_with_private_fields():
    gpu = open(os.path.join(ROOT, "bindings", "rust", "src", "gpu.rs")).read()
    for needle in ("impl Scatter for Lambertian", "impl Scatter for Metal", "impl Scatter for Dialectric",
                   "impl Hit for Sphere", "pub fn to_rt(&self) -> rt_camera", "pub fn flatten(", "UNCOMPILED"):
        assert needle in gpu, needle
