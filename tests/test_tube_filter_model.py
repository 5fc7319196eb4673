"""A numpy model of the shipped scan filter's device half (rt_device.hpp, make_tube / tube_a_words), run
against the real host tables (rt_tube_tile_host): every f32 operation is emulated with numpy.float32, the
hardware's approximate rsq/rcp are perturbed by up to 1 ulp either way, operands are cut to the bf16 pieces
the matrix pipe sees, and the products are summed exactly.  What is left to the device test
(tests/test_gpu_filter.py) is only the matrix pipe's own accumulation error, budgeted 33 u (|c| + |o|):
this test checks `the reference can hit  =>  both |H_k| < 2` (H = sigma h, sigma = 2 (1 - 2^-6) / bound folded into
the columns) WITH that budget still to spare."""
import numpy as np

import rtiow_amd as rt

U = 2.0 ** -24
f32 = np.float32


def bf16_rne(x):
    """float32 array -> float32 array holding the nearest bfloat16 (round to nearest even)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)) << np.uint32(16)
    return r.view(np.float32)


def two_pieces(x):
    p1 = bf16_rne(x)
    p2 = bf16_rne((x - p1).astype(f32))
    return p1.astype(np.float64) + p2.astype(np.float64)


def ulp_jitter(x, rng):
    """x times (1 + k 2^-23), k in {-1, 0, 1}: a 1-ulp-accurate hardware approximation."""
    return (x * (f32(1.0) + rng.integers(-1, 2, x.shape).astype(f32) * f32(2.0 ** -23))).astype(f32)


def model_rows(o, d, rho, rng):
    of, df = o.astype(f32), d.astype(f32)
    a = (df[:, 2] * df[:, 2] + (df[:, 1] * df[:, 1] + df[:, 0] * df[:, 0])).astype(f32)
    o1 = ((np.abs(of[:, 0]) + np.abs(of[:, 1])).astype(f32) + np.abs(of[:, 2])).astype(f32)      # |o|_1 >= |o| (rt_device.hpp, make_tube)
    s = ulp_jitter((f32(1.0) / np.sqrt(a.astype(np.float64))).astype(f32), rng)
    g = (df * s[:, None]).astype(f32)
    sg = np.where(np.signbit(g[:, 2]), f32(-1.0), f32(1.0)).astype(f32)
    aa = (-ulp_jitter((f32(1.0) / (sg + g[:, 2]).astype(np.float64)).astype(f32), rng)).astype(f32)
    b = ((g[:, 0] * g[:, 1]).astype(f32) * aa).astype(f32)
    e = (f32(128 * U) * o1 + f32(rho)).astype(f32)
    lam = (f32(rho) * ulp_jitter((f32(1.0) / e.astype(np.float64)).astype(f32), rng)).astype(f32)
    u = np.empty((len(o), 2, 3), dtype=f32)
    u[:, 0, 0] = lam * ((sg * g[:, 0]).astype(f32) * (g[:, 0] * aa).astype(f32) + f32(1.0)).astype(f32)
    u[:, 0, 1] = lam * (sg * b).astype(f32)
    u[:, 0, 2] = lam * (-sg * g[:, 0]).astype(f32)
    u[:, 1, 0] = lam * b
    u[:, 1, 1] = lam * (g[:, 1] * (g[:, 1] * aa).astype(f32) + sg).astype(f32)
    u[:, 1, 2] = lam * (-g[:, 1])
    t = -(u[:, :, 2] * of[:, None, 2] + (u[:, :, 1] * of[:, None, 1] + (u[:, :, 0] * of[:, None, 0]).astype(f32)).astype(f32)).astype(f32)
    return u, t, lam


def sigma_from_words(words):
    """(32,) f64: the bf16 scale factor of each column (K-slots 12..14)."""
    bits = np.array([words[32 + col][2] & 0xFFFF for col in range(32)], dtype=np.uint32)
    return (bits << np.uint32(16)).view(np.float32).astype(np.float64)


def columns_from_words(words):
    """(32, 3) f64: sigma x centre as the matrix pipe sees it, y1 + y2 of the host table."""
    c = np.empty((32, 3))
    for col in range(32):
        k0, k1 = words[col], words[32 + col]
        for i, w in enumerate((k0[0], k0[2], k1[0])):
            y1 = (np.uint32(w & 0xFFFF) << np.uint32(16)).view(np.float32)
            y2 = (np.uint32(w >> 16) << np.uint32(16)).view(np.float32)
            c[col, i] = float(y1) + float(y2)
    return c


def test_model_of_the_tube_filter_is_sound_with_the_accumulation_budget_to_spare():
    rng = np.random.default_rng(31)
    worst_basis = 0.0
    hits = kept = total = 0
    for it in range(30):
        sp = np.zeros(32, dtype=rt.SPHERE_DTYPE)
        sp["center"] = rng.uniform(-40, 40, (32, 3)); sp["center"][:, 1] = 0.2
        sp["radius"] = rng.choice([0.2, 0.2, 0.2, 1.0, 0.05], 32)
        words, bound, rho = rt.tube_tile_host(sp)
        c, r = sp["center"].astype(np.float64), sp["radius"].astype(np.float64)
        o = rng.uniform(-40, 40, (256, 3)); o[:, 1] = np.abs(o[:, 1]) * 0.1
        if it % 4 == 3:
            o[:64] *= 30.0                                   # far out on the ground sphere
        d = rng.standard_normal((256, 3)) * 10.0 ** rng.uniform(-3, 3, (256, 1))
        # aim three quarters of the rays at (or just past the rim of) a sphere: hits and grazing misses
        for k in range(192):
            j = k % 32
            to_c = c[j] - o[k]
            dist = np.linalg.norm(to_c)
            axis = np.cross(to_c, rng.standard_normal(3)); axis /= np.linalg.norm(axis)
            off = r[j] * rng.choice([0.0, 0.5, 1.0 - 1e-6, 1.0 + 1e-6])
            ang = np.arcsin(min(1.0, off / dist))
            d[k] = (np.cos(ang) * to_c / dist + np.sin(ang) * axis) * 10.0 ** rng.uniform(-2, 2)
        u, t, lam = model_rows(o, d, rho, rng)
        dn = d / np.linalg.norm(d, axis=1)[:, None]
        worst_basis = max(worst_basis, float(np.max(np.abs((u.astype(np.float64) * dn[:, None, :]).sum(2)) / lam[:, None]) / U),
                          float(np.max(np.abs(np.linalg.norm(u.astype(np.float64), axis=2) / lam[:, None] - 1.0)) / U))
        # what the matrix pipe multiplies: two bf16 pieces of every u component, the host's two pieces of c
        u2 = np.stack([two_pieces(u[:, k, i]) for k in range(2) for i in range(3)], axis=1).reshape(len(o), 2, 3)
        cw = columns_from_words(words)
        sigma = sigma_from_words(words)
        # exact sum of what the matrix pipe multiplies: H = (u pieces) . (sigma c pieces) + t sigma
        h = (u2[:, None, :, :] * cw[None, :, None, :]).sum(3) + t.astype(np.float64)[:, None, :] * sigma[None, :, None]
        oc = o[:, None, :] - c[None, :, :]
        hb = (oc * d[:, None, :]).sum(2)
        disc = hb ** 2 - (d ** 2).sum(1)[:, None] * ((oc ** 2).sum(2) - (r ** 2)[None, :])
        spare = 33.0 * U * (np.linalg.norm(c, axis=1)[None, :] + np.linalg.norm(o, axis=1)[:, None]) * sigma[None, :]
        ok = np.max(np.abs(h), axis=2) + spare < 2.0 * (1.0 - 2.0 ** -7)      # the kernel keeps |H| < 2: margin for the f32 result
        assert not np.any((disc >= 0.0) & ~ok)
        hits += int((disc >= 0.0).sum()); kept += int((np.max(np.abs(h), axis=2) < 2.0).sum()); total += disc.size
    assert worst_basis < 64.0, worst_basis          # the allowance of the proof (measured on the device: < 4 u)
    assert hits > 2000 and kept < 0.2 * total       # the cases do exercise both outcomes
