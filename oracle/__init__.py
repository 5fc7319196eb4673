"""ctypes loader of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module (see oracle.h).  The product package rtiow_amd never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ORACLE_LIB: another build of the same sources (tests/test_sanitizers.py points it at the ASan/UBSan build)
LIB_PATH = os.environ.get("ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")


class sphere(C.Structure):
    _fields_ = [("center", C.c_double * 3), ("radius", C.c_double), ("albedo", C.c_double * 3),
                ("param", C.c_double), ("kind", C.c_int32), ("reserved", C.c_int32)]


class camera(C.Structure):
    _fields_ = [("origin", C.c_double * 3), ("lower_left_corner", C.c_double * 3),
                ("horizontal", C.c_double * 3), ("vertical", C.c_double * 3),
                ("u", C.c_double * 3), ("v", C.c_double * 3), ("lens_radius", C.c_double)]


class params(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("spp", C.c_int32),
                ("sample_begin", C.c_int32), ("max_depth", C.c_int32),
                ("row_begin", C.c_int32), ("row_end", C.c_int32), ("row_step", C.c_int32),
                ("seed", C.c_uint64), ("t_min", C.c_double), ("nthreads", C.c_int32), ("flags", C.c_uint32)]


class stats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("rays_traced", C.c_uint64), ("depth_hist", C.c_uint64 * 64),
                ("end_sky", C.c_uint64), ("end_absorb", C.c_uint64), ("end_depth", C.c_uint64),
                ("seconds", C.c_double), ("threads_used", C.c_int32)]


_lib = None


def build():
    subprocess.run(["make", "-s", "-C", _HERE, os.path.basename(LIB_PATH)], check=True)


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    lib = C.CDLL(LIB_PATH)
    d3 = C.POINTER(C.c_double)
    lib.oracle_philox4x32_10.argtypes = [C.POINTER(C.c_uint32)] * 3
    lib.oracle_camera_new.argtypes = [d3, d3, d3, C.c_double, C.c_double, C.c_double, C.c_double, C.POINTER(camera)]
    lib.oracle_a_render.argtypes = [C.POINTER(camera), C.POINTER(sphere), C.c_int32, C.POINTER(params), C.c_void_p, C.POINTER(stats)]
    lib.oracle_b_render.argtypes = [C.POINTER(camera), C.POINTER(sphere), C.c_int32, C.POINTER(params), C.c_void_p, C.c_void_p, C.POINTER(stats)]
    lib.oracle_a_resolve_rgba8.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_void_p]
    lib.oracle_b_resolve_rgba8.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_void_p]
    lib.oracle_b_fix_to_f32.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    lib.oracle_b_quantize.argtypes = [C.c_double]
    lib.oracle_b_quantize.restype = C.c_uint64
    lib.oracle_sphere_hit.argtypes = [d3, C.c_double, d3, d3, C.c_double, C.c_double, d3, d3, d3, C.POINTER(C.c_int)]
    lib.oracle_world_hit.argtypes = [C.POINTER(sphere), C.c_int32, d3, d3, C.c_double, d3]
    lib.oracle_reflect.argtypes = [d3, d3, d3]
    lib.oracle_refract.argtypes = [d3, d3, C.c_double, d3]
    lib.oracle_reflectance.argtypes = [C.c_double, C.c_double]
    lib.oracle_reflectance.restype = C.c_double
    lib.oracle_scatter.argtypes = [C.POINTER(sphere), d3, d3, d3, C.c_int, d3, C.c_int, C.POINTER(C.c_int), d3, d3]
    lib.oracle_to_rgba.argtypes = [d3, C.c_int64, C.POINTER(C.c_uint8)]
    lib.oracle_get_ray.argtypes = [C.POINTER(camera), C.c_double, C.c_double, C.c_double, C.c_double, d3, d3]
    lib.oracle_uniforms.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, C.c_void_p]
    lib.oracle_hardware_threads.restype = C.c_int
    _lib = lib
    return lib


# ---- convenience wrappers (numpy in / numpy out) --------------------------------

def _d3(v):
    return (C.c_double * 3)(*[float(x) for x in v])


def philox(ctr, key):
    o = (C.c_uint32 * 4)()
    load().oracle_philox4x32_10((C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), o)
    return tuple(int(x) for x in o)


def camera_new(look_from, look_at, v_up, v_fov, aspect_ratio, aperture, focus_dist):
    c = camera()
    load().oracle_camera_new(_d3(look_from), _d3(look_at), _d3(v_up), v_fov, aspect_ratio, aperture, focus_dist, C.byref(c))
    return c


def book1_camera(width, height):
    """main.rs:108-118 with aspect = width/height."""
    return camera_new((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, float(width) / float(height), 0.1, 10.0)


FLAG_UNIFORM53 = 0x8


def make_params(width, height, spp, *, sample_begin=0, max_depth=50, rows=None, seed=1, t_min=1e-4, nthreads=0, uniform53=False):
    """rows = (begin, end, step) over image rows j (0 = bottom); default all rows.
    uniform53: two Philox words per uniform (53 random bits), the mirror of RT_FLAG_UNIFORM53."""
    p = params()
    p.width, p.height, p.spp, p.sample_begin, p.max_depth = width, height, spp, sample_begin, max_depth
    b, e, s = rows if rows is not None else (0, height, 1)
    p.row_begin, p.row_end, p.row_step = b, e, s
    p.seed, p.t_min, p.nthreads = seed, t_min, nthreads
    p.flags = FLAG_UNIFORM53 if uniform53 else 0
    return p


def n_rows(p):
    step = max(1, p.row_step)
    return max(0, (p.row_end - p.row_begin + step - 1) // step)


def _spheres_ptr(flat):
    flat = np.ascontiguousarray(flat)
    assert flat.dtype.itemsize == 72
    return flat, flat.ctypes.data_as(C.POINTER(sphere)), int(flat.shape[0])


def stats_dict(st):
    d = {k: getattr(st, k) for k, _ in stats._fields_ if k != "depth_hist"}
    d["depth_hist"] = [int(x) for x in st.depth_hist]
    return d


def render_a(cam, flat_spheres, p):
    """Oracle A (literal): returns (sum f64 [rows,W,3], stats)."""
    flat, ptr, n = _spheres_ptr(flat_spheres)
    out = np.zeros((n_rows(p), p.width, 3), dtype=np.float64)
    st = stats()
    rc = load().oracle_a_render(C.byref(cam), ptr, n, C.byref(p), out.ctypes.data_as(C.c_void_p), C.byref(st))
    assert rc == 0, rc
    return out, stats_dict(st)


def render_b(cam, flat_spheres, p):
    """Oracle B (kernel contract): returns (fix u64 [rows,W,3], sum f32 [rows,W,3], stats)."""
    flat, ptr, n = _spheres_ptr(flat_spheres)
    fix = np.zeros((n_rows(p), p.width, 3), dtype=np.uint64)
    sm = np.zeros((n_rows(p), p.width, 3), dtype=np.float32)
    st = stats()
    rc = load().oracle_b_render(C.byref(cam), ptr, n, C.byref(p), fix.ctypes.data_as(C.c_void_p),
                                sm.ctypes.data_as(C.c_void_p), C.byref(st))
    assert rc == 0, rc
    return fix, sm, stats_dict(st)


def resolve_a(sums, spp, flip=True):
    sums = np.ascontiguousarray(sums, dtype=np.float64)
    out = np.zeros((sums.shape[0], sums.shape[1], 4), dtype=np.uint8)
    load().oracle_a_resolve_rgba8(sums.ctypes.data_as(C.c_void_p), sums.shape[1], sums.shape[0], int(spp), int(flip),
                                  out.ctypes.data_as(C.c_void_p))
    return out


def resolve_b(fix, spp, flip=True):
    fix = np.ascontiguousarray(fix, dtype=np.uint64)
    out = np.zeros((fix.shape[0], fix.shape[1], 4), dtype=np.uint8)
    load().oracle_b_resolve_rgba8(fix.ctypes.data_as(C.c_void_p), fix.shape[1], fix.shape[0], int(spp), int(flip),
                                  out.ctypes.data_as(C.c_void_p))
    return out


def fix_to_f32(fix):
    fix = np.ascontiguousarray(fix, dtype=np.uint64)
    out = np.zeros(fix.shape, dtype=np.float32)
    load().oracle_b_fix_to_f32(fix.ctypes.data_as(C.c_void_p), fix.size, out.ctypes.data_as(C.c_void_p))
    return out


def camera_from_host(cam):
    """rtiow_amd.Camera (host mirror) -> oracle camera struct, field by field (no arithmetic)."""
    c = camera()
    for name in ("origin", "lower_left_corner", "horizontal", "vertical", "u", "v"):
        a = getattr(cam, name)
        setattr(c, name, (C.c_double * 3)(float(a[0]), float(a[1]), float(a[2])))
    c.lens_radius = float(cam.lens_radius)
    return c


def uniforms(seed, pixel, sample, count, uniform53=False, symmetric=False):
    """The first `count` draws of the stream of (pixel, sample), taken as one run: from [0,1), or (symmetric) from (-1..1)."""
    out = np.zeros(count, dtype=np.float64)
    flags = (FLAG_UNIFORM53 if uniform53 else 0) | (0x100 if symmetric else 0)
    load().oracle_uniforms(seed, pixel, sample, flags, count, out.ctypes.data_as(C.c_void_p))
    return out
