"""The default draws: ALL 32 bits of one Philox word -- u = w * 2^-32 for [0,1), x = (int32)w * 2^-31 for the symmetric ranges
(rounds 1-4: (w >> 8) * 2^-24 and 2u - 1) -- and the retry loops' rejection tests on them (vec3.rs:37-45, 59-68:
`length_squared() < 1.0`).

The reference evaluates x*x + y*y + z*z in f64 with a rounding per operation.  On the 2^-31 lattice of x the products are
no longer exact in f64 (m^2 has up to 62 bits), so the kernel's integer test S = sum m_i^2 < 2^62 is the reference's comparison
only away from the boundary; where the roundings could decide (the high dword of S is 2^30 - 1 or 2^30) the kernel evaluates the
f64 expression as written.  CPU: the error bound behind that window, on the oracle's arithmetic.  GPU: the kernel's own functions
(rt_unit_accept_device) against numpy's f64 evaluation of the reference's expression, on words chosen to land in and around the window."""
import numpy as np
import pytest

import rtiow_amd as rt


def f64_accepts(w):
    """The reference's test as written, in IEEE f64 without fused multiply-add (numpy): ((x*x + y*y) + z*z) < 1.0, x = (int32)w * 2^-31."""
    x = np.ascontiguousarray(w).view(np.int32).astype(np.float64) * (1.0 / 2147483648.0)
    xx = x * x
    return ((xx[:, 0] + xx[:, 1]) + xx[:, 2]) < 1.0


def exact_S(w):
    m = np.ascontiguousarray(w).view(np.int32).astype(object)
    return m[:, 0] * m[:, 0] + m[:, 1] * m[:, 1] + m[:, 2] * m[:, 2]


def isqrt64(v):
    """floor(sqrt(v)) for int64 arrays 0 <= v < 2^63."""
    r = np.sqrt(v.astype(np.float64)).astype(np.int64)
    r -= (r * r > v)
    r += ((r + 1) * (r + 1) <= v)
    return r


def words_near_the_boundary(rng, n, spread):
    """n triples of words whose exact sum of squares S = sum m_i^2, m_i = (int32)w_i, lies within about `spread` lattice units (2^-62 each)
    of 2^62, on both sides.  Coarse (spread >= 2^33): x, y random, z solved for, x nudged.  Fine: y random and z solved for among
    4 096 candidates so that y^2 + z^2 = 2^62 - r with a small r, then a SMALL x (|x| <= ~2^10: S moves by < 2^11 per step of x)
    with x^2 next to r."""
    if spread >= 2 ** 33:
        x = rng.integers(-2 ** 31 + 2 ** 20, 2 ** 31 - 2 ** 20, n)
        y = rng.integers(-2 ** 31, 2 ** 31, n)
        x, y = x[np.abs(x) > 2 ** 24], y[np.abs(x) > 2 ** 24]
        rest = 2 ** 62 - (x.astype(object) ** 2 + y.astype(object) ** 2)
        keep = np.array([r > 2 ** 40 for r in rest])
        x, y, rest = x[keep], y[keep], rest[keep].astype(np.int64)
        z = isqrt64(rest) + rng.integers(-2, 3, len(x))
        target = rng.integers(-spread, spread + 1, len(x))
        S = x.astype(object) ** 2 + y.astype(object) ** 2 + z.astype(object) ** 2
        x = x + np.array([int(round((2 ** 62 + int(t) - int(s_)) / (2.0 * int(xx)))) for t, s_, xx in zip(target, S, x)], dtype=np.int64)
        m = np.stack([x, y, z], axis=1)
    else:
        K = 4096
        y = rng.integers(2 ** 29, 2 ** 31 - 2 ** 20, (n, K))
        rest = np.int64(2 ** 62) - y * y
        z = isqrt64(rest)
        r = rest - z * z                                                    # y^2 + z^2 = 2^62 - r, 0 <= r <= 2 z
        k = np.argmin(r, axis=1)
        y, z, r = y[np.arange(n), k], z[np.arange(n), k], r[np.arange(n), k]
        x = isqrt64(r) + rng.integers(-1, 3, n)                             # x^2 - r in about (-2^11, 2^11)
        m = np.stack([x, y, z], axis=1)
        m = m * rng.choice([-1, 1], (n, 3))
        perm = rng.permuted(np.tile(np.arange(3), (n, 1)), axis=1)
        m = np.take_along_axis(m, perm, axis=1)
    m = m[np.all((m >= -2 ** 31) & (m < 2 ** 31), axis=1)]
    return np.ascontiguousarray(m.astype(np.int32)).view(np.uint32)


def test_the_window_covers_every_case_where_the_f64_roundings_decide():
    """Outside the window (high dword of S not 2^30 - 1 or 2^30) the f64 comparison IS the exact one: checked on ~66 000 triples
    piled up within 2^34 lattice units of the boundary, 6 000 of them within ~2^12."""
    rng = np.random.default_rng(5)
    w = np.concatenate([words_near_the_boundary(rng, 60000, 2 ** 34), words_near_the_boundary(rng, 6000, 2 ** 12)])
    S = exact_S(w)
    hi = np.array([int(s) >> 32 for s in S])
    inside = (hi == 2 ** 30 - 1) | (hi == 2 ** 30)
    assert inside.sum() > 10000 and (~inside).sum() > 10000                 # both sides are populated
    assert np.mean([abs(int(v) - 2 ** 62) < 2 ** 12 for v in S[-5000:]]) > 0.5    # (the fine generator does land next to the boundary)
    f = f64_accepts(w)
    exact = np.array([int(s) < 2 ** 62 for s in S])
    assert np.array_equal(f[~inside], exact[~inside])                       # away from the boundary: no rounding can decide
    # ... and inside it the roundings DO decide now and then (which is why the kernel cannot use the integer test alone there)
    near = np.array([abs(int(s) - 2 ** 62) < 2 ** 11 for s in S])
    assert (f[near] != exact[near]).any()
    # the stated error bound: |f64 sum - exact| < 2^-49
    x = w.view(np.int32).astype(np.float64) / 2.0 ** 31
    s64 = (x[:, 0] * x[:, 0] + x[:, 1] * x[:, 1]) + x[:, 2] * x[:, 2]
    from fractions import Fraction
    worst = max(abs(Fraction(float(a)) - Fraction(int(b), 2 ** 62)) for a, b in zip(s64[-2000:], S[-2000:]))
    assert worst < Fraction(1, 2 ** 49)


def test_oracle_takes_all_32_bits(oracle_mod):
    seed, pixel, sample = 99, 4242, 3
    words = []
    for e in range(2):
        words += list(oracle_mod.philox((pixel, sample, e, 0), (seed, 0)))
    u = oracle_mod.uniforms(seed, pixel, sample, 8)
    assert np.array_equal(u, np.array(words, dtype=np.float64) / 2.0 ** 32)
    assert any(w & 0xFF for w in words)                                     # (the low byte is in use: not the 24-bit rule)
    x = oracle_mod.uniforms(seed, pixel, sample, 8, symmetric=True)         # (-1..1): the word as a two's-complement integer
    assert np.array_equal(x, np.array(words, dtype=np.uint32).view(np.int32).astype(np.float64) / 2.0 ** 31)
    assert ((x >= -1.0) & (x < 1.0)).all() and (x < 0).any() and (x > 0).any()
    # with 53 bits the symmetric ranges stay 2u - 1
    u53 = oracle_mod.uniforms(seed, pixel, sample, 4, uniform53=True)
    assert np.array_equal(oracle_mod.uniforms(seed, pixel, sample, 4, uniform53=True, symmetric=True), 2.0 * u53 - 1.0)


@pytest.mark.gpu
def test_device_rejection_tests_are_the_references_f64_comparison(renderer):
    rng = np.random.default_rng(11)
    w = np.concatenate([
        rng.integers(0, 2 ** 32, (50000, 3), dtype=np.uint64).astype(np.uint32),        # anywhere
        words_near_the_boundary(rng, 50000, 2 ** 34),                                   # around the window's edges
        words_near_the_boundary(rng, 6000, 2 ** 12),                                    # where the roundings decide
        np.array([[0, 0, 0], [2 ** 31, 2 ** 31, 2 ** 31], [2 ** 32 - 1] * 3, [2 ** 31, 0, 0], [2 ** 31 - 1, 0, 0],
                  [0, 2 ** 31, 0], [2 ** 31 + 1, 0, 0]], dtype=np.uint32),
    ])
    acc, uni = renderer.unit_accept(w)
    assert np.array_equal((acc & 1).astype(bool), f64_accepts(w))
    disk = w.copy()
    disk[:, 2] = 0                                                          # z = 0.0: vec3.rs:65's Vec3::new(x, y, 0.0)
    assert np.array_equal((acc & 2).astype(bool), f64_accepts(disk))
    # a disk sample sits on ITS boundary for other words than a sphere sample: pairs (y, z) with y^2 + z^2 within 2^12 of 2^62
    # are rare (one in ~2^20 values of y): scan 2^23 consecutive y from a random start, on both sides of the circle
    y = np.arange(2 ** 23, dtype=np.int64) + int(rng.integers(2 ** 29, 2 ** 30))
    rest = np.int64(2 ** 62) - y * y
    z = isqrt64(rest)
    below = rest - z * z                                                    # y^2 + z^2 = 2^62 - below
    above = (z + 1) * (z + 1) - rest                                        # y^2 + (z+1)^2 = 2^62 + above
    pairs = np.concatenate([np.stack([y, z], 1)[below < 2 ** 12], np.stack([y, z + 1], 1)[above < 2 ** 12]])
    assert len(pairs) >= 4
    wd = np.zeros((len(pairs), 3), dtype=np.int64)
    wd[:, :2] = pairs * rng.choice([-1, 1], pairs.shape)
    wd = np.ascontiguousarray(wd.astype(np.int32)).view(np.uint32)
    acc_d, _ = renderer.unit_accept(wd)
    assert np.array_equal((acc_d & 2).astype(bool), f64_accepts(wd))
    assert np.array_equal((acc_d & 1).astype(bool), f64_accepts(wd))       # (z = 0.0: the sphere test of the same words agrees)
    # word -> draw: u = w * 2^-32; symmetric ranges (int32)w * 2^-31, exactly
    assert np.array_equal(uni[:, 0], w[:, 0].astype(np.float64) / 2.0 ** 32)
    for k in range(3):
        assert np.array_equal(uni[:, 1 + k], np.ascontiguousarray(w[:, k]).view(np.int32).astype(np.float64) / 2.0 ** 31)
