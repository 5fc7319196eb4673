"""The scan filter's matrix-pipe arithmetic against exact arithmetic.

DESIGN.md section 5.2 proves `filter drops a sphere => the reference misses it` from a bound
on the rounding error of the two K = 4 products.  The bound's only hardware assumption is
about the bf16 MFMA: products of bf16 pieces are exact and their 32-term sum is accumulated
with at most 64 u relative error per unit of sum(|terms|) (u = 2^-24).  These tests measure
that on the device, and check the end-to-end inequality on rays and spheres drawn like the
book-1 scene (ground sphere included).
"""
import numpy as np
import pytest

import rtiow_amd as rt

pytestmark = pytest.mark.gpu
U = 2.0 ** -24


def random_case(rng, scale_c=12.0):
    o = rng.uniform(-15, 15, (64, 3)); o[:, 1] = np.abs(o[:, 1]) * 0.2
    d = rng.standard_normal((64, 3)) * 10.0 ** rng.uniform(-3, 3, (64, 1))
    c = rng.uniform(-scale_c, scale_c, (16, 3)); c[:, 1] = 0.2
    r = np.full(16, 0.2)
    c[0] = (0.0, -1000.0, 0.0); r[0] = 1000.0          # the ground
    c[1] = (4.0, 1.0, 0.0); r[1] = 1.0
    return o, d, c, r


@pytest.mark.parametrize("bf16x3,KU,budget", [(True, 1024 * U, 472.0), (False, 128 * U, 61.0)])
def test_matrix_filter_products_within_the_proved_budget(renderer, bf16x3, KU, budget):
    rng = np.random.default_rng(11)
    worst = 0.0
    for _ in range(40):
        o, d, c, r = random_case(rng)
        # the per-ray rows exactly as make_filter() builds them (f32)
        of, df, cf = o.astype(np.float32), d.astype(np.float32), c.astype(np.float32)
        a = (df.astype(np.float64) ** 2).sum(1)
        g = (df / np.sqrt(a * (1.0 - KU))[:, None]).astype(np.float32)
        h0 = (of.astype(np.float64) * g.astype(np.float64)).sum(1).astype(np.float32)
        o2 = ((of.astype(np.float64) ** 2).sum(1) * (1.0 - KU / (1.0 - KU))).astype(np.float32)
        r1 = np.concatenate([-g, h0[:, None]], axis=1)
        r2 = np.concatenate([-2.0 * of, o2[:, None]], axis=1)
        s = np.concatenate([cf, np.ones((16, 1), np.float32)], axis=1)
        hb, q = renderer.filter_products(r1, r2, s, bf16x3=bf16x3)
        # exact values of the same bilinear forms on the SAME f32 operands
        hb_x = r1.astype(np.float64) @ s.astype(np.float64).T
        q_x = r2.astype(np.float64) @ s.astype(np.float64).T
        S = (o ** 2).sum(1)[:, None] + (c ** 2).sum(1)[None, :] + (r ** 2)[None, :]
        D = hb.astype(np.float64) ** 2 - q.astype(np.float64)
        D_x = hb_x ** 2 - q_x
        worst = max(worst, float(np.max(np.abs(D - D_x) / (U * S))))
        # and the conclusion itself: exact disc >= 0  =>  D'' >= K'
        kappa = KU / (1.0 - KU)
        kp = (c ** 2).sum(1) * (1.0 - kappa) - r ** 2 * (1.0 + 2.0 * kappa)
        oc = o[:, None, :] - c[None, :, :]
        hbt = (oc * d[:, None, :]).sum(2)
        disc = hbt ** 2 - (d ** 2).sum(1)[:, None] * ((oc ** 2).sum(2) - (r ** 2)[None, :])
        Df = (hb * hb - q).astype(np.float32)       # f32, like the kernel's fma (one more rounding)
        assert not np.any((disc >= 0.0) & (Df < kp[None, :].astype(np.float32)))
    # accumulation + operand errors only (inputs identical on both sides): far below the budget,
    # which also has to cover the roundings of o, c, g themselves
    assert worst < budget, worst
    print(f"bf16x3={bf16x3}: worst |D''-exact| = {worst:.2f} u S (budget {budget})")


def test_bf16x3_split_is_exact(renderer):
    """x == x1 + x2 + x3: with S = identity columns the product returns the operand itself."""
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((64, 4)) * 10.0 ** rng.uniform(-6, 6, (64, 4))).astype(np.float32)
    s = np.zeros((16, 4), np.float32)
    s[0, 0] = s[1, 1] = s[2, 2] = s[3, 3] = 1.0
    hb, q = renderer.filter_products(x, x, s, bf16x3=True)
    assert np.array_equal(hb[:, :4], x) and np.array_equal(q[:, :4], x)


# ---- the shipped scan mode: the filter as ONE contraction of 11 terms (rt_device.hpp) ----------

def _spheres16(c, r):
    sp = np.zeros(16, dtype=rt.SPHERE_DTYPE)
    sp["center"] = c
    sp["radius"] = r
    sp["albedo"] = 0.5
    return sp


def _lifted_exact(o, d, c, r, KU):
    """The real-number value the contraction approximates, and the reference's discriminant / a."""
    a = (d ** 2).sum(1)
    g = d / np.sqrt(a * (1.0 - KU))[:, None]
    kappa = KU / (1.0 - KU)
    oc = o[:, None, :] - c[None, :, :]
    hb = (oc * g[:, None, :]).sum(2)
    lhs = hb ** 2 - (oc ** 2).sum(2) + (r ** 2)[None, :] \
        + kappa * ((o ** 2).sum(1)[:, None] + (c ** 2).sum(1)[None, :] + 2.0 * (r ** 2)[None, :])
    hbt = (oc * d[:, None, :]).sum(2)
    disc = hbt ** 2 - a[:, None] * ((oc ** 2).sum(2) - (r ** 2)[None, :])
    return lhs, disc


def grazing_case(rng):
    """Rays aimed at the rim of the spheres: disc / a within a few 1e-7 of zero, both signs."""
    o, _, c, r = random_case(rng)
    d = np.empty((64, 3))
    for k in range(64):
        j = 2 + k % 14
        to_c = c[j] - o[k]
        dist = np.linalg.norm(to_c)
        axis = np.cross(to_c, rng.standard_normal(3)); axis /= np.linalg.norm(axis)
        off = r[j] * (1.0 + rng.uniform(-3e-6, 3e-6))          # miss distance ~ r
        ang = np.arcsin(min(1.0, off / dist))
        w = to_c / dist
        d[k] = (np.cos(ang) * w + np.sin(ang) * axis) * 10.0 ** rng.uniform(-2, 2)
    return o, d, c, r


def test_lifted_filter_within_the_proved_budget(renderer):
    KU = 1024 * U
    rng = np.random.default_rng(5)
    worst_eval = 0.0
    kept = total = hits = 0
    for it in range(60):
        o, d, c, r = grazing_case(rng) if it % 3 == 2 else random_case(rng)
        D, R, C = renderer.filter_lifted(o, d, _spheres16(c, r))
        assert np.all(R[:, 10] == 1.0)                         # every ray inside the analysed range
        S = (o ** 2).sum(1)[:, None] + (c ** 2).sum(1)[None, :] + (r ** 2)[None, :]
        # (1) the matrix pipe against exact arithmetic on the SAME f32 terms: dropped piece
        #     products (2.01 u) + accumulation, budgeted 391 u S
        Rx = R.astype(np.float64).copy(); Rx[:, 10] = 1.0
        exact_terms = Rx @ C.astype(np.float64).T
        worst_eval = max(worst_eval, float(np.max(np.abs(D - exact_terms) / (U * S))))
        # (2) the whole chain (operand roundings included) against the real-number identity:
        #     the kernel's value may fall short of it by less than the slack kappa S
        lhs, disc = _lifted_exact(o, d, c, r, KU)
        assert np.all(D.astype(np.float64) >= lhs - 460.0 * U * S), float(np.max((lhs - D) / (U * S)))
        # (3) the conclusion itself: the reference can hit  =>  kept
        assert not np.any((disc >= 0.0) & (D < 0.0))
        kept += int((D >= 0.0).sum()); total += D.size; hits += int((disc >= 0.0).sum())
    assert worst_eval < 391.0, worst_eval
    print(f"lifted: worst |D - exact sum of terms| = {worst_eval:.2f} u S (budget 391); kept {kept} of {total}, "
          f"reference can hit {hits}")


def test_lifted_columns_outside_the_analysed_range_are_always_kept(renderer):
    rng = np.random.default_rng(9)
    o, d, c, r = random_case(rng)
    c[3] = (1e16, 0.0, 0.0)              # |c|^2 + r^2 >= 1e30
    r[4] = 1e-16                         # r^2 <= 1e-30
    sp = _spheres16(c, r)
    D, R, C = renderer.filter_lifted(o, d, sp)
    assert np.all(D[:, 3] >= 0.0) and np.all(D[:, 4] >= 0.0)
    # and rays outside it are flagged for the exhaustive exact scan
    o2, d2 = o.copy(), d.copy()
    d2[0] = (1e-11, 0.0, 0.0); d2[1] = (1e11, 0.0, 0.0); o2[2] = (1e16, 0.0, 0.0)
    D, R, C = renderer.filter_lifted(o2, d2, sp)
    assert list(R[:3, 10]) == [0.0, 0.0, 0.0] and np.all(R[3:, 10] == 1.0)
    assert np.all(np.isfinite(D))


# ---- the shipped scan mode: the tube filter (rt_device.hpp, MODE 5) -------------------------------

def _spheres32(c, r):
    sp = np.zeros(32, dtype=rt.SPHERE_DTYPE)
    sp["center"] = c
    sp["radius"] = r
    sp["albedo"] = 0.5
    return sp


def tube_case(rng, grazing):
    o1, d1, c1, r1 = (grazing_case if grazing else random_case)(rng)
    _, _, c2, r2 = random_case(rng, scale_c=60.0)          # a second half with far-away centres
    c2[0] = c2[5] + 0.3; r2[0] = 0.05; r2[1] = 2.5         # (no second ground): mixed radii around the floor
    return o1, d1, np.concatenate([c1, c2]), np.concatenate([r1, r2])


def test_tube_filter_is_sound_and_within_its_budget(renderer):
    rng = np.random.default_rng(21)
    worst_eval = worst_basis = worst_norm = 0.0
    kept = total = hits = 0
    for it in range(60):
        o, d, c, r = tube_case(rng, grazing=(it % 3 == 2))
        if it % 5 == 4:
            o[:16] *= 40.0                                   # rays that start far out on the ground sphere
        h, rows, bound, rho = renderer.filter_tube(o, d, _spheres32(c, r))
        assert np.all(rows[:, 8] == 1.0)
        u = rows[:, :6].reshape(64, 2, 3).astype(np.float64)
        t = rows[:, 6:8].astype(np.float64)
        dn = d / np.linalg.norm(d, axis=1)[:, None]
        on = np.linalg.norm(o, axis=1)
        lam = rho / (rho + 128.0 * U * on)
        # (1) the basis: lambda u_k is perpendicular to the ray and of length lambda, to well inside 64 u
        worst_basis = max(worst_basis, float(np.max(np.abs((u * dn[:, None, :]).sum(2)) / lam[:, None]) / U))
        worst_norm = max(worst_norm, float(np.max(np.abs(np.linalg.norm(u, axis=2) / lam[:, None] - 1.0)) / U))
        # (2) the matrix pipe against exact arithmetic on the same rows and the exact centres:
        #     operand truncation (2^-16 per factor) + accumulation, budgeted (513 + 33) u |c| + 33 u |o|
        hx = (u[:, None, :, :] * c[None, :, None, :]).sum(3) + t[:, None, :]
        cn = np.linalg.norm(c, axis=1)
        scale = U * (546.0 * cn[None, :, None] + 33.0 * on[:, None, None])
        worst_eval = max(worst_eval, float(np.max(np.abs(h - hx) / scale)))
        # (3) the conclusion: the reference can hit  =>  both |h_k| <= bound
        oc = o[:, None, :] - c[None, :, :]
        hbt = (oc * d[:, None, :]).sum(2)
        disc = hbt ** 2 - (d ** 2).sum(1)[:, None] * ((oc ** 2).sum(2) - (r ** 2)[None, :])
        keep = np.max(np.abs(h), axis=2) <= bound[None, :]
        assert not np.any((disc >= 0.0) & ~keep)
        kept += int(keep.sum()); total += keep.size; hits += int((disc >= 0.0).sum())
    assert worst_basis < 64.0 and worst_norm < 64.0, (worst_basis, worst_norm)
    assert worst_eval < 1.0, worst_eval
    print(f"tube: |u.d|/lambda <= {worst_basis:.1f} u, ||u|/lambda - 1| <= {worst_norm:.1f} u (allowed 64 u); "
          f"evaluation error {worst_eval:.3f} of its budget; kept {kept} of {total}, reference can hit {hits}")


def test_tube_columns_and_rays_outside_the_analysed_range(renderer):
    rng = np.random.default_rng(23)
    o, d, c, r = tube_case(rng, grazing=False)
    c[3] = (1e16, 0.0, 0.0)              # |c|^2 + r^2 >= 1e30: always kept
    r[4] = 1e-16                         # r^2 <= 1e-30: always kept
    h, rows, bound, rho = renderer.filter_tube(o, d, _spheres32(c, r))
    assert np.isinf(bound[3]) and np.isinf(bound[4]) and np.all(np.isfinite(h))
    assert np.all(bound[np.isfinite(bound)] >= rho)
    o2, d2 = o.copy(), d.copy()
    d2[0] = (1e-11, 0.0, 0.0); d2[1] = (1e11, 0.0, 0.0); o2[2] = (1e16, 0.0, 0.0)
    h, rows, bound, rho = renderer.filter_tube(o2, d2, _spheres32(c, r))
    assert list(rows[:3, 8]) == [0.0, 0.0, 0.0] and np.all(rows[3:, 8] == 1.0)
    # a ray outside the range is never "kept" by the matrix test (it is tested exhaustively instead)
    assert np.all(np.max(np.abs(h[:3]), axis=2)[:, np.isfinite(bound)] > bound[np.isfinite(bound)][None, :])
