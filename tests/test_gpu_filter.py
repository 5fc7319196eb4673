"""The shipped scan filter (the tube filter, scan mode 5): its matrix-pipe arithmetic against exact
arithmetic, measured on the device, and the end-to-end inequality
`hit => kept` on random and rim-grazing rays.  The earlier matrix forms (scan modes 2-4) have the
same kind of tests in test_gpu_crosscheck_modes.py, against the cross-check build."""
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
        sp32 = _spheres32(c, r)
        H, rows, bound, rho = renderer.filter_tube(o, d, sp32)
        words, _, _ = rt.tube_tile_host(sp32)
        sigma = (np.array([words[32 + col][2] & 0xFFFF for col in range(32)], dtype=np.uint32) << np.uint32(16)).view(np.float32).astype(np.float64)
        assert np.all(sigma > 0) and np.all(sigma * bound <= 2.0 * (1.0 - 2.0 ** -6) * (1.0 + 1e-6))
        h = H / sigma[None, :, None]                        # back to length units
        assert np.all(rows[:, 8] == 1.0)
        u = rows[:, :6].reshape(64, 2, 3).astype(np.float64)
        t = rows[:, 6:8].astype(np.float64)
        dn = d / np.linalg.norm(d, axis=1)[:, None]
        on = np.linalg.norm(o, axis=1)
        lam = rho / (rho + 128.0 * U * np.abs(o).sum(1))          # e = 128 u |o|_1 (the 1-norm bounds the 2-norm: rt_device.hpp, make_tube)
        # (1) the basis: lambda u_k is perpendicular to the ray and of length lambda, to well inside 64 u
        worst_basis = max(worst_basis, float(np.max(np.abs((u * dn[:, None, :]).sum(2)) / lam[:, None]) / U))
        worst_norm = max(worst_norm, float(np.max(np.abs(np.linalg.norm(u, axis=2) / lam[:, None] - 1.0)) / U))
        # (2) the matrix pipe against exact arithmetic on the same rows and the exact centres:
        #     operand truncation (2^-16 per factor) + accumulation, budgeted (513 + 33) u |c| + 33 u |o|
        hx = (u[:, None, :, :] * c[None, :, None, :]).sum(3) + t[:, None, :]
        cn = np.linalg.norm(c, axis=1)
        scale = U * (546.0 * cn[None, :, None] + 33.0 * on[:, None, None])
        worst_eval = max(worst_eval, float(np.max(np.abs(h - hx) / scale)))
        # (3) the conclusion: the reference can hit  =>  the kernel keeps the pair: both |H_k| < 2
        oc = o[:, None, :] - c[None, :, :]
        hbt = (oc * d[:, None, :]).sum(2)
        disc = hbt ** 2 - (d ** 2).sum(1)[:, None] * ((oc ** 2).sum(2) - (r ** 2)[None, :])
        keep = np.max(np.abs(H), axis=2) < 2.0
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
    assert np.all(h[:, 3] == 0.0) and np.all(h[:, 4] == 0.0)            # always kept: sigma = 0, H = 0
    assert np.all(bound[np.isfinite(bound)] >= rho)
    o2, d2 = o.copy(), d.copy()
    d2[0] = (1e-11, 0.0, 0.0); d2[1] = (1e11, 0.0, 0.0); o2[2] = (1e16, 0.0, 0.0)
    h, rows, bound, rho = renderer.filter_tube(o2, d2, _spheres32(c, r))
    assert list(rows[:3, 8]) == [0.0, 0.0, 0.0] and np.all(rows[3:, 8] == 1.0)
    # a ray outside the range is never "kept" by the matrix test (it is tested exhaustively instead):
    # its t = 3e38 times sigma overflows or stays huge
    big = np.max(np.abs(h[:3]), axis=2)[:, np.isfinite(bound)]
    assert np.all(~(big < 2.0))
