"""Renderer: the host-side caller of the C ABI (stands in for main.rs:122-147).

Nothing here computes radiance: every pixel comes from librtiow_hip.so.
"""
import ctypes as C

import numpy as np

from . import _ffi
from .scene import SPHERE_DTYPE


def make_params(width, height, spp, *, sample_begin=0, max_depth=50, t_min=1e-4, seed=1,
                tile_rows=8, shard_index=0, shard_count=1, flags=0):
    p = _ffi.rt_params()
    p.width, p.height, p.spp, p.sample_begin = int(width), int(height), int(spp), int(sample_begin)
    p.max_depth, p.t_min, p.seed = int(max_depth), float(t_min), int(seed)
    p.tile_rows, p.shard_index, p.shard_count, p.flags = int(tile_rows), int(shard_index), int(shard_count), int(flags)
    return p


def shard_rows(params):
    lib = _ffi.load()
    rows = C.c_int32(0)
    _ffi.check(lib.rt_shard_rows(C.byref(params), C.byref(rows)), "rt_shard_rows")
    return rows.value


def shard_row_indices(params):
    """Image row j (0 = bottom) of every compact row of the shard."""
    lib = _ffi.load()
    n = shard_rows(params)
    out = np.empty(n, dtype=np.int32)
    j = C.c_int32(0)
    for r in range(n):
        _ffi.check(lib.rt_shard_row_index(C.byref(params), r, C.byref(j)), "rt_shard_row_index")
        out[r] = j.value
    return out


def stats_dict(st):
    out = {}
    for k, _ in _ffi.rt_stats._fields_:
        v = getattr(st, k)
        out[k] = list(v) if hasattr(v, "__len__") else v
    return out


def tile_layout_host(flat):
    """Where rt_upload_scene puts each sphere in the filter's table of columns (rt_tile_layout_host, no GPU):
    ((grid_dim, n_global), grid[8] f32, slot_of[columns] i32 with -1 for padding)."""
    lib = _ffi.load()
    flat = np.ascontiguousarray(flat, dtype=SPHERE_DTYPE)
    dims = (C.c_int32 * 2)()
    grid = (C.c_float * 8)()
    cap = 2 * len(flat) + (48 + 63 * 63) * 32 + 64
    slot = np.full(cap, -2, dtype=np.int32)
    n = lib.rt_tile_layout_host(flat.ctypes.data_as(C.POINTER(_ffi.rt_sphere)), len(flat), dims, grid,
                                slot.ctypes.data_as(C.POINTER(C.c_int32)), len(slot))
    if n < 0:
        _ffi.check(n, "rt_tile_layout_host")
    return (int(dims[0]), int(dims[1])), np.array(list(grid), dtype=np.float32), slot[:n].copy()


def tube_tile_host(spheres32):
    """Host-side half of one tube-filter tile (no GPU): words (64, 4) u32, bound (32,) f32, rho."""
    lib = _ffi.load()
    sp = np.ascontiguousarray(spheres32)
    assert sp.shape == (32,) and sp.dtype.itemsize == C.sizeof(_ffi.rt_sphere)
    words = np.zeros((64, 4), dtype=np.uint32)
    bound = np.zeros(32, dtype=np.float32)
    rho = C.c_float(0.0)
    _ffi.check(lib.rt_tube_tile_host(sp.ctypes.data_as(C.POINTER(_ffi.rt_sphere)), words.ctypes.data_as(C.c_void_p),
                                     bound.ctypes.data_as(C.c_void_p), C.byref(rho)), "rt_tube_tile_host")
    return words, bound, float(rho.value)


class Renderer:
    """One rt_context (one GPU)."""

    def __init__(self, device_id=0):
        self._lib = _ffi.load()
        h = C.c_void_p()
        _ffi.check(self._lib.rt_create(int(device_id), C.byref(h)), "rt_create")
        self._h = h
        self.n_spheres = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- scene ---------------------------------------------------------------
    def upload_scene(self, world):
        """world: HittableList, or a numpy array of SPHERE_DTYPE records."""
        flat = world.flatten() if hasattr(world, "flatten") and not isinstance(world, np.ndarray) else world
        flat = np.ascontiguousarray(flat, dtype=SPHERE_DTYPE)
        ptr = flat.ctypes.data_as(C.POINTER(_ffi.rt_sphere))
        _ffi.check(self._lib.rt_upload_scene(self._h, ptr, int(flat.shape[0])), "rt_upload_scene")
        self.n_spheres = int(flat.shape[0])

    # -- host-buffer render --------------------------------------------------
    def render(self, cam, params, want_fix=True):
        """Returns (sum f32 [rows,W,3], fix u64 [rows,W,3] or None, stats dict)."""
        rc = cam.to_rt_camera() if hasattr(cam, "to_rt_camera") else cam
        rows = shard_rows(params)
        out_sum = np.zeros((rows, params.width, 3), dtype=np.float32)
        out_fix = np.zeros((rows, params.width, 3), dtype=np.uint64) if want_fix else None
        st = _ffi.rt_stats()
        _ffi.check(self._lib.rt_render(self._h, C.byref(rc), C.byref(params),
                                       out_sum.ctypes.data_as(C.c_void_p),
                                       out_fix.ctypes.data_as(C.c_void_p) if want_fix else None,
                                       C.byref(st)), "rt_render")
        return out_sum, out_fix, stats_dict(st)

    # -- main.rs:122-145 in one call ---------------------------------------------
    def render_rgba8(self, cam, params, flip=True):
        """Render + Color::to_rgba + flip with the sums kept on the device (rt_render_rgba8): returns
        (RGBA8 [rows,W,4] -- the bytes ImageBuffer::from_vec takes at main.rs:147 --, stats dict)."""
        rc = cam.to_rt_camera() if hasattr(cam, "to_rt_camera") else cam
        rows = shard_rows(params)
        out = np.zeros((rows, params.width, 4), dtype=np.uint8)
        st = _ffi.rt_stats()
        _ffi.check(self._lib.rt_render_rgba8(self._h, C.byref(rc), C.byref(params), int(bool(flip)),
                                             out.ctypes.data_as(C.c_void_p), C.byref(st)), "rt_render_rgba8")
        return out, stats_dict(st)

    # -- device-buffer render (pointers come from e.g. torch tensors) ---------
    def render_device(self, cam, params, d_fix_ptr, stream=0):
        rc = cam.to_rt_camera() if hasattr(cam, "to_rt_camera") else cam
        _ffi.check(self._lib.rt_render_device(self._h, C.byref(rc), C.byref(params),
                                              C.c_void_p(d_fix_ptr), C.c_void_p(stream)), "rt_render_device")

    def fix_to_f32_device(self, d_fix_ptr, count, d_out_ptr, stream=0):
        _ffi.check(self._lib.rt_fix_to_f32_device(self._h, C.c_void_p(d_fix_ptr), int(count),
                                                  C.c_void_p(d_out_ptr), C.c_void_p(stream)), "rt_fix_to_f32_device")

    def resolve_rgba8_device(self, d_fix_ptr, width, rows, spp, flip, d_rgba_ptr, stream=0):
        _ffi.check(self._lib.rt_resolve_rgba8_device(self._h, C.c_void_p(d_fix_ptr), int(width), int(rows),
                                                     int(spp), int(flip), C.c_void_p(d_rgba_ptr),
                                                     C.c_void_p(stream)), "rt_resolve_rgba8_device")

    def last_stats(self):
        st = _ffi.rt_stats()
        _ffi.check(self._lib.rt_last_stats(self._h, C.byref(st)), "rt_last_stats")
        return stats_dict(st)

    # -- to_rgba + flip --------------------------------------------------------
    def resolve_rgba8(self, fix, spp, flip=True):
        """fix: exact sums u64 [rows,W,3] -> RGBA8 [rows,W,4] (vec3.rs:403-421, main.rs:141-145)."""
        fix = np.ascontiguousarray(fix, dtype=np.uint64)
        rows, width = fix.shape[0], fix.shape[1]
        out = np.zeros((rows, width, 4), dtype=np.uint8)
        _ffi.check(self._lib.rt_resolve_rgba8(self._h, fix.ctypes.data_as(C.c_void_p), width, rows,
                                              int(spp), int(bool(flip)), out.ctypes.data_as(C.c_void_p)),
                   "rt_resolve_rgba8")
        return out

    def f64_div_sqrt(self, a, b):
        """Device-side a/b and sqrt(a) in f64 (known-answer test hook)."""
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        q = np.zeros_like(a)
        r = np.zeros_like(a)
        _ffi.check(self._lib.rt_f64_div_sqrt_device(self._h, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p),
                                                    int(a.size), q.ctypes.data_as(C.c_void_p),
                                                    r.ctypes.data_as(C.c_void_p)), "rt_f64_div_sqrt_device")
        return q, r

    def quantize(self, x):
        """The kernel's own quantisation of radiance values to the 2^-32 grid (known-answer test hook)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        q = np.zeros(x.shape, dtype=np.uint64)
        _ffi.check(self._lib.rt_quantize_device(self._h, x.ctypes.data_as(C.c_void_p), int(x.size), q.ctypes.data_as(C.c_void_p)),
                   "rt_quantize_device")
        return q

    def unit_accept(self, words):
        """The kernel's own rejection tests and word -> uniform rules on (n, 3) Philox words (known-answer test hook):
        returns (accept u32 [n]: bit 0 unit sphere, bit 1 unit disk; uniforms f64 [n, 4] = u01(wx), u11(wx), u11(wy), u11(wz))."""
        w = np.ascontiguousarray(words, dtype=np.uint32).reshape(-1, 3)
        acc = np.zeros(len(w), dtype=np.uint32)
        uni = np.zeros((len(w), 4), dtype=np.float64)
        _ffi.check(self._lib.rt_unit_accept_device(self._h, w.ctypes.data_as(C.c_void_p), len(w), acc.ctypes.data_as(C.c_void_p),
                                                   uni.ctypes.data_as(C.c_void_p)), "rt_unit_accept_device")
        return acc, uni

    def filter_products(self, r1, r2, s, bf16x3=True):
        """Matrix-pipe filter products HB = R1 x S^T, Q = R2 x S^T of scan modes 2/3 (known-answer test
        hook; cross-check build only)."""
        if not _ffi.has_crosscheck_modes():
            raise _ffi.RtiowHipError("rt_filter_products_device needs the -DRTIOW_CROSSCHECK_MODES build (RTIOW_HIP_LIB)")
        r1 = np.ascontiguousarray(r1, dtype=np.float32).reshape(64, 4)
        r2 = np.ascontiguousarray(r2, dtype=np.float32).reshape(64, 4)
        s = np.ascontiguousarray(s, dtype=np.float32).reshape(16, 4)
        hb = np.zeros((64, 16), dtype=np.float32)
        q = np.zeros((64, 16), dtype=np.float32)
        _ffi.check(self._lib.rt_filter_products_device(self._h, r1.ctypes.data_as(C.c_void_p), r2.ctypes.data_as(C.c_void_p),
                                                       s.ctypes.data_as(C.c_void_p), int(bool(bf16x3)),
                                                       hb.ctypes.data_as(C.c_void_p), q.ctypes.data_as(C.c_void_p)),
                   "rt_filter_products_device")
        return hb, q

    def filter_lifted(self, o, d, spheres16):
        """One tile of the single-contraction filter (scan mode 4; cross-check build only), as the kernel evaluates it.

        o, d: (64, 3) f64 rays; spheres16: structured array (SPHERE_DTYPE) of 16 spheres.
        Returns D (64, 16) f32, R (64, 11) f32 per-ray terms, C (16, 11) f32 per-sphere terms."""
        if not _ffi.has_crosscheck_modes():
            raise _ffi.RtiowHipError("rt_filter_lifted_device needs the -DRTIOW_CROSSCHECK_MODES build (RTIOW_HIP_LIB)")
        o = np.ascontiguousarray(o, dtype=np.float64).reshape(64, 3)
        d = np.ascontiguousarray(d, dtype=np.float64).reshape(64, 3)
        sp = np.ascontiguousarray(spheres16)
        assert sp.shape == (16,) and sp.dtype.itemsize == C.sizeof(_ffi.rt_sphere)
        D = np.zeros((64, 16), dtype=np.float32)
        R = np.zeros((64, 11), dtype=np.float32)
        Cc = np.zeros((16, 11), dtype=np.float32)
        _ffi.check(self._lib.rt_filter_lifted_device(self._h, o.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p),
                                                     sp.ctypes.data_as(C.POINTER(_ffi.rt_sphere)),
                                                     D.ctypes.data_as(C.c_void_p), R.ctypes.data_as(C.c_void_p),
                                                     Cc.ctypes.data_as(C.c_void_p)), "rt_filter_lifted_device")
        return D, R, Cc

    def filter_tube(self, o, d, spheres32):
        """One tile of the tube filter (the shipped scan mode), as the kernel evaluates it.

        o, d: (64, 3) f64 rays; spheres32: structured array (SPHERE_DTYPE) of 32 spheres.
        Returns h (64, 32, 2) f32, rows (64, 9) f32, bound (32,) f32, rho."""
        o = np.ascontiguousarray(o, dtype=np.float64).reshape(64, 3)
        d = np.ascontiguousarray(d, dtype=np.float64).reshape(64, 3)
        sp = np.ascontiguousarray(spheres32)
        assert sp.shape == (32,) and sp.dtype.itemsize == C.sizeof(_ffi.rt_sphere)
        h = np.zeros((64, 32, 2), dtype=np.float32)
        rows = np.zeros((64, 9), dtype=np.float32)
        bound = np.zeros(32, dtype=np.float32)
        rho = C.c_float(0.0)
        _ffi.check(self._lib.rt_filter_tube_device(self._h, o.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p),
                                                   sp.ctypes.data_as(C.POINTER(_ffi.rt_sphere)),
                                                   h.ctypes.data_as(C.c_void_p), rows.ctypes.data_as(C.c_void_p),
                                                   bound.ctypes.data_as(C.c_void_p), C.byref(rho)), "rt_filter_tube_device")
        return h, rows, bound, float(rho.value)

    def philox(self, ctr, key):
        c = (C.c_uint32 * 4)(*ctr)
        k = (C.c_uint32 * 2)(*key)
        o = (C.c_uint32 * 4)()
        _ffi.check(self._lib.rt_philox_device(self._h, c, k, o), "rt_philox_device")
        return tuple(int(x) for x in o)
