"""Scan modes 2-4 (earlier matrix-pipe forms of the filter): compiled only into
tools/librtiow_hip_xcheck.so (-DRTIOW_CROSSCHECK_MODES, built by __graft_entry__.build()); the
product library does not carry them.  `test_crosscheck_build_in_a_subprocess` re-runs this file and
the bit-exact parity tests against that library, once per mode, in child processes
(RTIOW_HIP_LIB / RTIOW_SCAN_MODE are read when the library is loaded / a context is created).
The other tests here skip unless the loaded library is that build.

The error budgets measured here: products of bf16 pieces are exact and their 32-term sum is
accumulated with at most 64 u relative error per unit of sum(|terms|) (u = 2^-24).
"""
import os
import subprocess
import sys

import numpy as np
import pytest

import rtiow_amd as rt
from rtiow_amd import _ffi

pytestmark = pytest.mark.gpu
U = 2.0 ** -24
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
XCHECK_LIB = os.path.join(ROOT, "tools", "librtiow_hip_xcheck.so")


def needs_xcheck():
    if not _ffi.has_crosscheck_modes():
        pytest.skip("product library loaded: scan modes 2-4 live in tools/librtiow_hip_xcheck.so")


def test_crosscheck_build_in_a_subprocess():
    """The cross-check build: its known-answer tests, then bit-exact parity vs Oracle B under each of the
    scan modes 2, 3, 4 (and 1, 5 of the same build)."""
    if _ffi.has_crosscheck_modes():
        pytest.skip("already inside the cross-check run")
    assert os.path.exists(XCHECK_LIB), "tools/librtiow_hip_xcheck.so missing: run __graft_entry__.build()"
    base = dict(os.environ, RTIOW_HIP_LIB=XCHECK_LIB)
    runs = [("5", ["tests/test_gpu_crosscheck_modes.py", "tests/test_gpu_properties.py::test_all_scan_filters_give_the_same_bits"])]
    parity = ["tests/test_gpu_parity.py", "-k", "bit_exact or tie_rule or filter_never or tenk or tie_between"]
    runs += [(m, parity) for m in ("2", "3", "4", "1")]
    for mode, what in runs:
        r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", *what],
                           cwd=ROOT, env=dict(base, RTIOW_SCAN_MODE=mode), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, f"scan mode {mode}: {r.stdout[-2000:]}\n{r.stderr[-1000:]}"
        assert " passed" in r.stdout and "no tests ran" not in r.stdout


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
    needs_xcheck()
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
    needs_xcheck()
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
    needs_xcheck()
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
    needs_xcheck()
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

