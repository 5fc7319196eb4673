"""Build-authored analytic known answers for the oracle (the reference has no
tests).  Each case names the reference lines it pins."""
import ctypes as C
import math

import numpy as np
import pytest


def d3(*v):
    return (C.c_double * 3)(*[float(x) for x in v])


def hit(lib, c, r, o, d, t_min=1e-4, t_max=math.inf):
    t = C.c_double(); p = d3(0, 0, 0); n = d3(0, 0, 0); f = C.c_int(0)
    ok = lib.oracle_sphere_hit(d3(*c), float(r), d3(*o), d3(*d), t_min, t_max, C.byref(t), p, n, C.byref(f))
    return ok, t.value, list(p), list(n), f.value


def make_sphere(oracle_mod, c, r, kind=0, albedo=(0.5, 0.5, 0.5), param=0.0):
    s = oracle_mod.sphere()
    s.center = d3(*c); s.radius = r; s.albedo = d3(*albedo); s.param = param; s.kind = kind
    return s


def test_axis_aligned_roots(oracle_mod):            # sphere.rs:18-34
    lib = oracle_mod.load()
    ok, t, p, n, front = hit(lib, (0, 0, -5), 1.0, (0, 0, 0), (0, 0, -1))
    assert ok and t == 4.0 and p == [0, 0, -4] and n == [0, 0, 1] and front == 1
    # non-unit direction: t scales, a = |d|^2 is not assumed 1
    ok, t, p, n, front = hit(lib, (0, 0, -5), 1.0, (0, 0, 0), (0, 0, -2))
    assert ok and t == 2.0 and p == [0, 0, -4]


def test_inside_sphere_takes_far_root_and_flips_normal(oracle_mod):   # sphere.rs:29-33, mod.rs:21-22
    lib = oracle_mod.load()
    ok, t, p, n, front = hit(lib, (0, 0, 0), 2.0, (0, 0, 0), (1, 0, 0))
    assert ok and t == 2.0 and p == [2, 0, 0] and front == 0 and n == [-1, 0, 0]


def test_t_min_boundary(oracle_mod):                # `root < t_min` is strict: root == t_min is a hit
    lib = oracle_mod.load()
    ok, t, *_ = hit(lib, (0, 0, -2), 1.0, (0, 0, 0), (0, 0, -1), t_min=1.0)
    assert ok and t == 1.0
    ok, t, *_ = hit(lib, (0, 0, -2), 1.0, (0, 0, 0), (0, 0, -1), t_min=math.nextafter(1.0, 2.0))
    assert ok and t == 3.0                          # near root rejected, far root taken
    ok, *_ = hit(lib, (0, 0, -2), 1.0, (0, 0, 0), (0, 0, -1), t_min=3.5)
    assert not ok


def test_t_max_boundary(oracle_mod):                # `t_max < root` is strict: root == t_max is a hit
    lib = oracle_mod.load()
    assert hit(lib, (0, 0, -2), 1.0, (0, 0, 0), (0, 0, -1), t_max=1.0)[0]
    assert not hit(lib, (0, 0, -2), 1.0, (0, 0, 0), (0, 0, -1), t_max=math.nextafter(1.0, 0.0))[0]


def test_grazing_ray_disc_zero_is_a_hit(oracle_mod):    # sphere.rs:25: only disc < 0 misses
    lib = oracle_mod.load()
    ok, t, p, n, front = hit(lib, (0, 1, -3), 1.0, (0, 0, 0), (0, 0, -1))
    assert ok and t == 3.0 and p == [0, 0, -3]


def test_miss_and_behind(oracle_mod):
    lib = oracle_mod.load()
    assert not hit(lib, (0, 3, -5), 1.0, (0, 0, 0), (0, 0, -1))[0]
    assert not hit(lib, (0, 0, 5), 1.0, (0, 0, 0), (0, 0, -1))[0]


def test_tie_rule_later_sphere_wins(oracle_mod):    # mod.rs:61-67 + sphere.rs:29
    lib = oracle_mod.load()
    arr = (oracle_mod.sphere * 3)(make_sphere(oracle_mod, (0, 0, -5), 1.0),
                                  make_sphere(oracle_mod, (0, 0, -5), 1.0),
                                  make_sphere(oracle_mod, (0, 0, -9), 1.0))
    t = C.c_double()
    assert lib.oracle_world_hit(arr, 3, d3(0, 0, 0), d3(0, 0, -1), 1e-4, C.byref(t)) == 1 and t.value == 4.0
    assert lib.oracle_world_hit(arr, 1, d3(0, 0, 0), d3(0, 0, -1), 1e-4, C.byref(t)) == 0
    assert lib.oracle_world_hit(arr, 3, d3(0, 0, 0), d3(0, 1, 0), 1e-4, C.byref(t)) == -1


def test_reflect_refract_identities(oracle_mod):    # vec3.rs:116-125
    lib = oracle_mod.load()
    out = d3(0, 0, 0)
    lib.oracle_reflect(d3(1, -1, 0), d3(0, 1, 0), out)
    assert list(out) == [1, 1, 0]
    s = math.sqrt(0.5)
    lib.oracle_refract(d3(s, -s, 0), d3(0, 1, 0), 1.0, out)          # ratio 1: straight through
    assert np.allclose(list(out), [s, -s, 0], atol=1e-15)
    lib.oracle_refract(d3(0, -1, 0), d3(0, 1, 0), 1.0 / 1.5, out)    # normal incidence
    assert np.allclose(list(out), [0, -1, 0], atol=1e-15)
    lib.oracle_refract(d3(s, -s, 0), d3(0, 1, 0), 1.0 / 1.5, out)    # Snell: sin' = sin/1.5
    assert abs(out[0] - s / 1.5) < 1e-15 and abs(math.hypot(out[0], out[1]) - 1.0) < 1e-15


def test_schlick(oracle_mod):                       # materials.rs:78-82
    lib = oracle_mod.load()
    assert abs(lib.oracle_reflectance(1.0, 1.5) - 0.04) < 1e-16
    assert lib.oracle_reflectance(0.0, 1.5) == 1.0
    assert abs(lib.oracle_reflectance(1.0, 1.0 / 1.5) - 0.04) < 1e-15


def scatter(oracle_mod, mat, d_in, n, front, u):
    lib = oracle_mod.load()
    arr = (C.c_double * max(1, len(u)))(*u)
    used = C.c_int(0); att = d3(0, 0, 0); dout = d3(0, 0, 0)
    ok = lib.oracle_scatter(C.byref(mat), d3(*d_in), d3(0, 0, 0), d3(*n), front, arr, len(u), C.byref(used), att, dout)
    return ok, used.value, list(att), list(dout)


def test_lambertian_rejection_order_and_direction(oracle_mod):      # materials.rs:21-31, vec3.rs:37-49
    m = make_sphere(oracle_mod, (0, 0, 0), 1.0, kind=0, albedo=(0.1, 0.2, 0.3))
    # first triple (u=1-eps -> ~(1,1,1)) is outside the unit sphere and rejected; second accepted
    hi = 1.0 - 2.0 ** -24
    ok, used, att, dout = scatter(oracle_mod, m, (0, 0, -1), (0, 0, 1), 1, [hi, hi, hi, 0.5, 0.5, 0.75])
    assert ok and used == 6 and att == [0.1, 0.2, 0.3]
    assert dout == [0.0, 0.0, 2.0]                  # n + unit((0,0,0.5)); direction NOT normalised


def test_lambertian_near_zero_falls_back_to_normal(oracle_mod):     # vec3.rs:111-114
    m = make_sphere(oracle_mod, (0, 0, 0), 1.0, kind=0)
    ok, used, att, dout = scatter(oracle_mod, m, (0, 0, -1), (0, 0, 1), 1, [0.5, 0.5, 0.25])   # unit = (0,0,-1)
    assert ok and dout == [0.0, 0.0, 1.0]


def test_metal_draws_even_with_zero_fuzz_and_absorbs_below_surface(oracle_mod):   # materials.rs:48-62
    m = make_sphere(oracle_mod, (0, 0, 0), 1.0, kind=1, albedo=(0.7, 0.6, 0.5), param=0.0)
    s = math.sqrt(0.5)
    ok, used, att, dout = scatter(oracle_mod, m, (s, 0, -s), (0, 0, 1), 1, [0.5, 0.5, 0.75])
    assert ok and used == 3 and att == [0.7, 0.6, 0.5]
    assert np.allclose(dout, [s, 0, s], atol=1e-15)
    m2 = make_sphere(oracle_mod, (0, 0, 0), 1.0, kind=1, param=1.0)
    # grazing reflection pushed below the surface by the fuzz sample (0,0,-0.9): absorbed
    ok, used, *_ = scatter(oracle_mod, m2, (1, 0, -1e-3), (0, 0, 1), 1, [0.5, 0.5, 0.05])
    assert not ok and used == 3


def test_dialectric_draw_only_when_refraction_possible(oracle_mod):  # materials.rs:88-96
    m = make_sphere(oracle_mod, (0, 0, 0), 1.0, kind=2, param=1.5)
    # front face, normal incidence: can_refract; reflectance 0.04 <= 0.5 -> refract straight
    ok, used, att, dout = scatter(oracle_mod, m, (0, 0, -1), (0, 0, 1), 1, [0.5])
    assert ok and used == 1 and att == [1, 1, 1] and np.allclose(dout, [0, 0, -1], atol=1e-15)
    # u below the reflectance: reflect
    ok, used, att, dout = scatter(oracle_mod, m, (0, 0, -1), (0, 0, 1), 1, [0.01])
    assert used == 1 and np.allclose(dout, [0, 0, 1], atol=1e-15)
    # total internal reflection (inside, ratio 1.5, 60 degrees): NO draw, reflect
    d = (math.sin(math.radians(60)), 0, -math.cos(math.radians(60)))
    ok, used, att, dout = scatter(oracle_mod, m, d, (0, 0, 1), 0, [0.9])
    assert ok and used == 0 and np.allclose(dout, [d[0], 0, -d[2]], atol=1e-15)


def test_to_rgba_edges(oracle_mod):                 # vec3.rs:403-421
    lib = oracle_mod.load()
    out = (C.c_uint8 * 4)()
    lib.oracle_to_rgba(d3(0, 100.0, 25.0), 100, out)
    assert list(out) == [0, 255, 128, 255]          # sqrt(1)=1 -> clamp 0.999 -> 255 ; sqrt(.25)=.5 -> 128
    lib.oracle_to_rgba(d3(float("nan"), -1.0, 1e9), 1, out)
    assert list(out) == [0, 0, 255, 255]            # NaN -> 0 (`as u8`), sqrt(-1)=NaN -> 0
    lib.oracle_to_rgba(d3(0.999 ** 2, 0.5 ** 2, (255.0 / 256.0) ** 2), 1, out)
    assert out[1] == 128 and out[0] == 255


def test_quantize_contract_c5(oracle_mod):
    lib = oracle_mod.load()
    q = lib.oracle_b_quantize
    assert q(0.0) == 0 and q(-1.0) == 0 and q(float("nan")) == 0
    assert q(1.0) == 1 << 32 and q(0.5) == 1 << 31
    assert q(2.0 ** -33) == 0 and q(2.0 ** -32) == 1
    assert q(1e30) == 1 << 48 and q(65536.0) == 1 << 48 and q(65535.5) == (65535 << 32) + (1 << 31)      # clamp at 2^16 (RT_SAMPLE_CLAMP)
    assert q(1.0 - 2.0 ** -53) == (1 << 32) - 1     # truncation, not rounding


def test_get_ray_pinhole_centre(oracle_mod):        # camera.rs:47-54
    lib = oracle_mod.load()
    cam = oracle_mod.camera_new((0, 0, 0), (0, 0, -1), (0, 1, 0), 90.0, 1.0, 0.0, 1.0)
    o = d3(0, 0, 0); d = d3(0, 0, 0)
    lib.oracle_get_ray(C.byref(cam), 0.5, 0.5, 0.3, -0.2, o, d)
    assert list(o) == [0, 0, 0] and np.allclose(list(d), [0, 0, -1], atol=1e-15)
    lib.oracle_get_ray(C.byref(cam), 0.0, 0.0, 0.0, 0.0, o, d)
    assert np.allclose(list(d), [-1, -1, -1], atol=1e-15)        # lower-left corner at 90 degrees


def test_nan_and_infinite_roots_are_accepted_as_the_reference_accepts_them(oracle_mod):
    """sphere.rs:29-33 rejects with `root < t_min || t_max < root`: false for a NaN root, so a zero-length direction
    (a = 0, half_b = 0, disc = 0: root = 0/0) HITS every sphere; mod.rs:61-67 then carries closest_so_far = NaN, with which
    `t_max < root` is false too: every later sphere with a root >= t_min (or NaN) is accepted and the LAST one wins,
    whatever its distance.  An underflowing direction (a = 0, half_b != 0) gives roots of +-inf: +inf is accepted."""
    lib = oracle_mod.load()
    ok, t, p, n, front = hit(lib, (0, 0, -5), 1.0, (0, 0, 0), (0, 0, 0))
    assert ok and math.isnan(t) and all(math.isnan(v) for v in p)
    far, near = make_sphere(oracle_mod, (0, 0, -50), 1.0), make_sphere(oracle_mod, (0, 0, -5), 1.0)
    for order, want in (((near, far), 1), ((far, near), 1)):                   # zero direction: the last of the list, always
        arr = (oracle_mod.sphere * 2)(*order)
        t = C.c_double()
        assert lib.oracle_world_hit(arr, 2, d3(0, 0, 0), d3(0, 0, 0), 1e-4, C.byref(t)) == want and math.isnan(t.value)
    # (an ordinary ray through the same list is unaffected: the nearest root >= t_min wins)
    three = (oracle_mod.sphere * 3)(near, make_sphere(oracle_mod, (0, 0, 0), 1e-3), far)
    t = C.c_double()
    assert lib.oracle_world_hit(three, 3, d3(0, 0, 0), d3(0, 0, -1), 1e-4, C.byref(t)) == 1 and abs(t.value - 1e-3) < 1e-12
    assert lib.oracle_world_hit(three, 3, d3(0, 0, 0), d3(0, 0, 0), 1e-4, C.byref(t)) == 2 and math.isnan(t.value)
    # an underflowing direction: |d|^2 = 0 but d != 0; half_b^2 underflows too, disc = 0, root = -half_b / 0 = +inf
    tiny = 1e-170
    ok, t, p, n, front = hit(lib, (0, 0, -5), 1.0, (0, 0, 0), (0, 0, -tiny))
    assert ok and t == math.inf
    ok, t, p, n, front = hit(lib, (0, 0, 5), 1.0, (0, 0, 0), (0, 0, -tiny))   # moving away: -inf, then the far root -inf: a miss
    assert not ok
