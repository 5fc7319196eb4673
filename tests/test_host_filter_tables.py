"""The host half of the shipped scan filter (the tube filter), without a GPU: the
per-sphere columns and bounds rt_upload_scene builds, against a numpy restatement of what they must be."""
import numpy as np
import pytest

import rtiow_amd as rt

U = 2.0 ** -24


def bf16_to_f64(bits):
    return (bits.astype(np.uint32) << np.uint32(16)).view(np.float32).astype(np.float64)


def make_spheres(rng):
    sp = np.zeros(32, dtype=rt.SPHERE_DTYPE)
    sp["center"] = rng.uniform(-60, 60, (32, 3)) * 10.0 ** rng.uniform(-3, 1, (32, 1))
    sp["radius"] = rng.uniform(0.05, 2.5, 32)
    sp["albedo"] = 0.5
    return sp


def sigma_of(words):
    """(32,) f64: the bf16 scale factor of each column (K-slots 12..14 of the host table)."""
    return bf16_to_f64(np.array([words[32 + col][2] & 0xFFFF for col in range(32)], dtype=np.uint32))


def test_columns_are_two_exact_bf16_pieces_of_sigma_times_the_f64_centre():
    rng = np.random.default_rng(4)
    for _ in range(20):
        sp = make_spheres(rng)
        words, bound, rho = rt.tube_tile_host(sp)
        lo, hi = words & 0xFFFF, words >> 16
        sigma = sigma_of(words)
        # sigma = 2 (1 - 2^-6) / bound rounded DOWN to 8 significant bits: "hit => |H| <= sigma bound < 2"
        target = 2.0 * (1.0 - 2.0 ** -6) / bound.astype(np.float64)
        assert np.all(sigma <= target * (1.0 + 2.0 ** -23)) and np.all(sigma >= target * (1.0 - 2.0 ** -7))
        for col in range(32):
            c = sp["center"][col] * sigma[col]
            k0, k1 = words[col], words[32 + col]           # K-slots 0..7 and 8..15 of this column
            for i, (w_a, w_b) in enumerate([(k0[0], k0[1]), (k0[2], k0[3]), (k1[0], k1[1])]):
                assert w_a == w_b                           # (y1, y2, y1, y2)
                y1, y2 = bf16_to_f64(np.uint32(w_a & 0xFFFF)), bf16_to_f64(np.uint32(w_a >> 16))
                assert abs(c[i] - (y1 + y2)) <= 2.0 ** -16 * abs(c[i])
            sg = int(k1[2] & 0xFFFF)
            assert k1[2] == (sg | (sg << 16)) and k1[3] == sg       # (sigma, sigma, sigma, 0) against the pieces of t and the 1
        assert lo.shape == hi.shape == (64, 4)


def test_bounds_cover_the_error_budget_and_round_up():
    rng = np.random.default_rng(5)
    for _ in range(20):
        sp = make_spheres(rng)
        words, bound, rho = rt.tube_tile_host(sp)
        r = np.abs(sp["radius"])
        cn = np.linalg.norm(sp["center"], axis=1)
        need = r * (1.0 + 64 * U) + 640 * U * cn
        assert rho == np.float32(np.sort(r)[len(r) // 4])                   # lower quartile of the radii
        assert np.all(bound.astype(np.float64) >= np.maximum(need, rho))    # never below the proven bound
        assert np.all(bound.astype(np.float64) <= np.maximum(need, rho) * (1.0 + 2.0 ** -22))   # and tight


def test_out_of_range_spheres_are_always_kept_and_carry_no_centre():
    rng = np.random.default_rng(6)
    sp = make_spheres(rng)
    sp["center"][3] = (1e16, 0.0, 0.0)          # |c|^2 + r^2 >= 1e30
    sp["radius"][4] = 1e-16                     # r^2 <= 1e-30
    words, bound, rho = rt.tube_tile_host(sp)
    assert np.isinf(bound[3]) and np.isinf(bound[4])
    for col in (3, 4):                          # sigma = 0 and no centre: H = 0 for every ray, which is "kept"
        assert not words[col].any() and not words[32 + col].any()
