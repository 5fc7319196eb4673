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
