"""The C-ABI shared library: it loads, exports every symbol include/rtiow_hip.h
declares, validates arguments, and has NO CPU fallback (no compute calls here)."""
import ctypes as C
import os
import re

import pytest

from rtiow_amd import _ffi
import rtiow_amd as rt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions(crosscheck=False):
    src = open(os.path.join(ROOT, "include", "rtiow_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    # declarations under RTIOW_CROSSCHECK_MODES belong to the cross-check build only
    xsrc = "".join(re.findall(r"#ifdef RTIOW_CROSSCHECK_MODES(.*?)#endif", src, flags=re.S))
    src = re.sub(r"#ifdef RTIOW_CROSSCHECK_MODES.*?#endif", "", src, flags=re.S)
    names = lambda t: sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", t)))
    return (names(xsrc) if crosscheck else names(src))


def test_library_exports_every_declared_symbol():
    lib = _ffi.load()
    names = header_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/rtiow_hip.h but not exported"
    assert sorted(n for n, _, _ in _ffi.SYMBOLS) == names      # the binding covers the whole header
    xnames = header_functions(crosscheck=True)
    assert sorted(n for n, _, _ in _ffi.XCHECK_SYMBOLS) == xnames and len(xnames) == 2
    if "RTIOW_HIP_LIB" not in os.environ:                      # the product library does not carry modes 2-4
        assert not any(hasattr(lib, n) for n in xnames) and not _ffi.has_crosscheck_modes()


def test_crosscheck_library_exports_the_whole_header():
    """tools/librtiow_hip_xcheck.so (built by __graft_entry__.build()) = the product ABI + the two hooks."""
    path = os.path.join(ROOT, "tools", "librtiow_hip_xcheck.so")
    if not os.path.exists(path):
        pytest.skip("cross-check library not built")
    lib = C.CDLL(path)
    for n in header_functions() + header_functions(crosscheck=True):
        assert hasattr(lib, n), n


def test_struct_layouts_match_the_header():
    assert C.sizeof(_ffi.rt_sphere) == 72 and C.sizeof(_ffi.rt_camera) == 152
    assert C.sizeof(_ffi.rt_params) == 56 and _ffi.rt_params.t_min.offset == 24 and _ffi.rt_params.seed.offset == 32
    assert rt.SPHERE_DTYPE.itemsize == 72
    assert rt.SPHERE_DTYPE.fields["kind"][1] == _ffi.rt_sphere.kind.offset == 64


def test_identity():
    lib = _ffi.load()
    assert lib.rt_backend_name() == b"hip-gfx950"
    assert lib.rt_abi_version() == 5
    # the library on disk was built from the kernel sources of this tree (a stale .so would carry another sha)
    import bench
    assert lib.rt_build_source_sha().decode() == bench.kernel_source_sha(), "stale rtiow_amd/librtiow_hip.so: run ./build_lib.sh"


@pytest.mark.parametrize("kw,msg", [
    (dict(width=1), "width and height"), (dict(height=1), "width and height"),
    (dict(spp=-1), "spp"), (dict(t_min=0.0), "t_min"), (dict(tile_rows=0), "tile_rows"),
    (dict(shard_index=2, shard_count=2), "shard_index"), (dict(shard_count=0), "shard_index"),
    (dict(max_depth=-1), "max_depth"), (dict(flags=0x20), "unknown flags 0x20"), (dict(flags=0x80000001), "unknown flags"),
])
def test_parameter_validation(kw, msg):
    base = dict(width=8, height=8, spp=1)
    base.update(kw)
    p = rt.make_params(base.pop("width"), base.pop("height"), base.pop("spp"), **base)
    rows = C.c_int32()
    rc = _ffi.load().rt_shard_rows(C.byref(p), C.byref(rows))
    assert rc == -1 and msg in _ffi.load().rt_last_error().decode()


def test_null_arguments_are_errors_not_crashes():
    lib = _ffi.load()
    assert lib.rt_shard_rows(None, None) == -1
    assert lib.rt_create(0, None) == -1
    assert lib.rt_destroy(None) == 0
    assert lib.rt_upload_scene(None, None, 0) == -1
    assert lib.rt_render_rgba8(None, None, None, 1, None, None) == -1


def _gpu_present():
    return os.path.exists("/dev/kfd")


@pytest.mark.skipif(_gpu_present(), reason="this check is for machines without a GPU")
def test_no_cpu_fallback_without_a_gpu():
    """On a box with no HIP device the product must fail loudly, not fall back."""
    with pytest.raises(rt.RtiowHipError) as e:
        rt.Renderer(0)
    assert "no HIP device" in str(e.value) or "fallback" in str(e.value)


def test_product_library_carries_no_crosscheck_kernels():
    """Scan modes 2-4 live in rtiow_amd/csrc/xcheck/ and are compiled only into tools/librtiow_hip_xcheck.so: the product library's
    device code holds render_kernel<0|1|5, ...> instantiations only (the kernel names are in the .so's embedded code object)."""
    import re
    blob = open(_ffi.LIB_PATH if "RTIOW_HIP_LIB" not in os.environ else os.path.join(ROOT, "rtiow_amd", "librtiow_hip.so"), "rb").read()
    modes = set(int(m) for m in re.findall(rb"_ZN2rt13render_kernelILi(\d)E", blob))
    assert modes == {0, 1, 5}, modes
    x = os.path.join(ROOT, "tools", "librtiow_hip_xcheck.so")
    if os.path.exists(x):
        xmodes = set(int(m) for m in re.findall(rb"_ZN2rt13render_kernelILi(\d)E", open(x, "rb").read()))
        assert xmodes == {0, 1, 2, 3, 4, 5}, xmodes
    # ... and the product sources hold none of their definitions (only the include stubs under RTIOW_CROSSCHECK_MODES)
    for rel in ("rt_kernels.hpp", "rt_device.hpp", "rt_api.hip"):
        src = open(os.path.join(ROOT, "rtiow_amd", "csrc", rel)).read()
        assert "LiftedRay make_lifted" not in src and "a_operand_bf16x3(float" not in src and "auto check_sign_half" not in src, rel
