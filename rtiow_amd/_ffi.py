"""ctypes binding of librtiow_hip.so (include/rtiow_hip.h).

The library is the only compute backend: there is no Python or CPU fallback.
Loading fails loudly when the shared object has not been built
(`python -c "import __graft_entry__ as g; g.build()"` or ./build_lib.sh).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RTIOW_HIP_LIB: load another build of the same ABI instead (the tests use it for the cross-check
# build tools/librtiow_hip_xcheck.so, which also carries scan modes 2-4 and their known-answer hooks)
LIB_PATH = os.environ.get("RTIOW_HIP_LIB") or os.path.join(_HERE, "librtiow_hip.so")


class rt_sphere(C.Structure):
    _fields_ = [("center", C.c_double * 3), ("radius", C.c_double), ("albedo", C.c_double * 3),
                ("param", C.c_double), ("kind", C.c_int32), ("reserved", C.c_int32)]


class rt_camera(C.Structure):
    _fields_ = [("origin", C.c_double * 3), ("lower_left_corner", C.c_double * 3),
                ("horizontal", C.c_double * 3), ("vertical", C.c_double * 3),
                ("u", C.c_double * 3), ("v", C.c_double * 3), ("lens_radius", C.c_double)]


class rt_params(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("spp", C.c_int32),
                ("sample_begin", C.c_int32), ("max_depth", C.c_int32), ("t_min", C.c_double),
                ("seed", C.c_uint64), ("tile_rows", C.c_int32), ("shard_index", C.c_int32),
                ("shard_count", C.c_int32), ("flags", C.c_uint32)]


class rt_stats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("rays_traced", C.c_uint64), ("sphere_tests", C.c_uint64),
                ("candidates", C.c_uint64), ("exact_roots", C.c_uint64), ("kernel_ms", C.c_float), ("n_spheres", C.c_int32),
                ("grid_blocks", C.c_int32), ("block_threads", C.c_int32), ("scan_mode", C.c_int32), ("kernel_variant", C.c_int32),
                ("live_per_bounce", C.c_uint64 * 64), ("direct_samples", C.c_uint64)]


RT_FLAG_ACCUMULATE = 0x1
RT_FLAG_NO_FILTER = 0x2
RT_FLAG_DIAG_STATS = 0x4
RT_FLAG_UNIFORM53 = 0x8
RT_FLAG_OVERLAPPED = 0x10
RT_FLAG_KNOWN = 0x1f

# every symbol include/rtiow_hip.h declares: (name, restype, argtypes)
_VP = C.c_void_p
SYMBOLS = [
    ("rt_create", C.c_int, [C.c_int32, C.POINTER(_VP)]),
    ("rt_destroy", C.c_int, [_VP]),
    ("rt_upload_scene", C.c_int, [_VP, C.POINTER(rt_sphere), C.c_int32]),
    ("rt_shard_rows", C.c_int, [C.POINTER(rt_params), C.POINTER(C.c_int32)]),
    ("rt_shard_row_index", C.c_int, [C.POINTER(rt_params), C.c_int32, C.POINTER(C.c_int32)]),
    ("rt_render", C.c_int, [_VP, C.POINTER(rt_camera), C.POINTER(rt_params), _VP, _VP, C.POINTER(rt_stats)]),
    ("rt_render_device", C.c_int, [_VP, C.POINTER(rt_camera), C.POINTER(rt_params), _VP, _VP]),
    ("rt_fix_to_f32_device", C.c_int, [_VP, _VP, C.c_int64, _VP, _VP]),
    ("rt_last_stats", C.c_int, [_VP, C.POINTER(rt_stats)]),
    ("rt_resolve_rgba8_device", C.c_int, [_VP, _VP, C.c_int32, C.c_int32, C.c_int64, C.c_int32, _VP, _VP]),
    ("rt_resolve_rgba8", C.c_int, [_VP, _VP, C.c_int32, C.c_int32, C.c_int64, C.c_int32, _VP]),
    ("rt_render_rgba8", C.c_int, [_VP, C.POINTER(rt_camera), C.POINTER(rt_params), C.c_int32, _VP, C.POINTER(rt_stats)]),
    ("rt_last_error", C.c_char_p, []),
    ("rt_backend_name", C.c_char_p, []),
    ("rt_abi_version", C.c_int32, []),
    ("rt_build_source_sha", C.c_char_p, []),
    ("rt_philox_device", C.c_int, [_VP, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("rt_f64_div_sqrt_device", C.c_int, [_VP, _VP, _VP, C.c_int32, _VP, _VP]),
    ("rt_quantize_device", C.c_int, [_VP, _VP, C.c_int32, _VP]),
    ("rt_unit_accept_device", C.c_int, [_VP, _VP, C.c_int32, _VP, _VP]),
    ("rt_tube_tile_host", C.c_int, [C.POINTER(rt_sphere), _VP, _VP, C.POINTER(C.c_float)]),
    ("rt_tile_layout_host", C.c_int, [C.POINTER(rt_sphere), C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_int32), C.c_int32]),
    ("rt_filter_tube_device", C.c_int, [_VP, _VP, _VP, C.POINTER(rt_sphere), _VP, _VP, _VP, C.POINTER(C.c_float)]),
]

# declared under RTIOW_CROSSCHECK_MODES in the header: present only in the cross-check build
XCHECK_SYMBOLS = [
    ("rt_filter_products_device", C.c_int, [_VP, _VP, _VP, _VP, C.c_int32, _VP, _VP]),
    ("rt_filter_lifted_device", C.c_int, [_VP, _VP, _VP, C.POINTER(rt_sphere), _VP, _VP, _VP]),
]

_lib = None


class RtiowHipError(RuntimeError):
    pass


def load():
    """Returns the loaded library with prototypes set; raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RtiowHipError(
            f"{LIB_PATH} not found: build it with ./build_lib.sh (hipcc --offload-arch=gfx950); "
            "there is no CPU fallback for the render path")
    # One process, one HIP runtime.  PyTorch ships its own libamdhip64.so.7 and the dynamic
    # loader binds every later request for that SONAME to whichever copy came first; PyTorch
    # only works on its own copy, this library works on either.  So when torch is installed
    # it is imported first (device memory, streams and torch.distributed come from it anyway).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    for name, res, args in XCHECK_SYMBOLS:
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
    _lib = lib
    return lib


def has_crosscheck_modes():
    """True when the loaded library is the -DRTIOW_CROSSCHECK_MODES build (scan modes 2-4 and their hooks)."""
    return hasattr(load(), "rt_filter_lifted_device")


def check(rc, what):
    if rc != 0:
        msg = load().rt_last_error().decode("utf-8", "replace")
        raise RtiowHipError(f"{what} failed ({rc}): {msg}")
