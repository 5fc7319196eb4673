"""The shipped filter tiles its spheres by position and a wave scans only the tiles its rays can reach (DESIGN.md
section 5.2).  CPU tests of the two halves of that:

  * the host half, through the real library (rt_tile_layout_host, no device needed): every sphere that goes through the
    filter sits in exactly one column, and what the kernel assumes of a cell's tile holds (centre in the cell, radius
    within the pad, extent in y within the slab);
  * the device half (rt_device.hpp, grid_cells) as a numpy.float32 model with the hardware's approximate reciprocals
    perturbed by an ulp either way, run against those tables: whenever the reference's own f64 arithmetic finds a root
    with t >= t_min for a sphere of a cell tile, that cell is in the ray's footprint -- with the kernel's margins reduced
    to a QUARTER, so the shipped ones have room to spare.
"""
import ctypes as C

import numpy as np
import pytest

import rtiow_amd as rt
from rtiow_amd import _ffi

f32 = np.float32


def layout(flat):
    (g, ng), grid, slot = rt.tile_layout_host(flat)
    return g, ng, grid, slot


def spheres(centers, radii):
    flat = np.zeros(len(radii), dtype=rt.SPHERE_DTYPE)
    flat["center"] = centers
    flat["radius"] = radii
    flat["albedo"] = 0.5
    return flat


def scene_cases():
    rng = np.random.default_rng(11)
    book = rt.random_scene(1).flatten()
    tenk = rt.random_scene(1, grid=(-50, 49)).flatten()
    # spheres at random heights: a thick slab
    n = 900
    cloud = spheres(rng.uniform(-20, 20, (n, 3)), rng.uniform(0.1, 0.5, n))
    # two dense clusters and a sparse rest: cells overflow into the global tiles
    c = np.concatenate([rng.normal((-5, 0.3, -5), 0.6, (300, 3)), rng.normal((6, 0.3, 4), 0.4, (250, 3)), rng.uniform(-15, 15, (150, 3)) * (1, 0.02, 1)])
    clusters = spheres(c, rng.uniform(0.05, 0.2, len(c)))
    # all centres on one line in x (the z extent is zero), mixed sizes, a few huge ones
    line = spheres(np.stack([np.linspace(-30, 30, 400), np.full(400, 0.2), np.zeros(400)], 1), np.where(np.arange(400) % 50 == 0, 3.0, 0.2))
    # every centre the same point
    same = spheres(np.tile([[1.0, 2.0, 3.0]], (100, 1)), np.linspace(0.1, 0.3, 100))
    small = rt.random_scene(1, grid=(-3, 3)).flatten()       # <= 64 spheres go through the filter: no grid
    # legal coordinates (|x| < 1e15, rt_upload_scene) whose xz extent is beyond what f32 cell arithmetic can carry: two
    # groups around x = -9e14 and x = +9e14 (extent 1.8e15) -- the grid must be OFF, not a box that misses most spheres
    wide = spheres(np.concatenate([rng.uniform(-5, 5, (100, 3)) + (-9e14, 0, 0), rng.uniform(-5, 5, (100, 3)) + (9e14, 0, 0)]),
                   np.full(200, 0.2))
    # one axis huge (x spans 1.6e14, inside the limit), the other ten units wide: the grid stays on and must still hold its spheres
    huge_x = spheres(np.stack([rng.uniform(-8e13, 8e13, 300), rng.uniform(0, 1, 300), rng.uniform(-5, 5, 300)], 1), np.full(300, 0.2))
    # both axes huge but inside the limit: a real G x G grid at the edge of what the f32 cell arithmetic carries
    huge_xz = spheres(np.stack([rng.uniform(-8e13, 8e13, 2000), rng.uniform(0, 1, 2000), rng.uniform(-8e13, 8e13, 2000)], 1), np.full(2000, 0.2))
    return {"book": book, "tenk": tenk, "cloud": cloud, "clusters": clusters, "line": line, "same": same, "small": small,
            "wide": wide, "huge_x": huge_x, "huge_xz": huge_xz}


CASES = scene_cases()


@pytest.mark.parametrize("name", list(CASES))
def test_every_filtered_sphere_has_one_column_and_cells_hold_what_the_kernel_assumes(name):
    flat = CASES[name]
    G, n_global, g, slot_of = layout(flat)
    n = len(flat)
    assert len(slot_of) % 32 == 0 and len(slot_of) <= 65536
    used = slot_of[slot_of >= 0]
    assert len(np.unique(used)) == len(used) and used.max() < n
    # what is in no column is the always-exact list: at most 8 spheres, each more than 8x the median radius
    missing = np.setdiff1d(np.arange(n), used)
    r = np.abs(flat["radius"])
    assert len(missing) <= 8 and np.all(r[missing] > 8 * np.median(r))
    if name in ("small", "wide"):
        assert G == 0
    if name == "same":
        assert G == 1
    if G == 0:
        assert np.array_equal(used, np.sort(used)) and np.all(slot_of[used] == used)      # columns in list order
        return
    assert name in ("same", "line", "huge_x") or G > 1        # (a 1-D distribution of spheres: one cell can be the cheapest grid)
    assert len(slot_of) == 32 * (n_global + G * G) and 0 <= n_global <= 48
    x0, z0, inv, x1, z1, ylo, yhi, pad = [float(v) for v in g]
    for t in range(n_global, n_global + G * G):
        idx = slot_of[32 * t:32 * t + 32]
        idx = idx[idx >= 0]
        if len(idx) == 0:
            continue
        ix, iz = (t - n_global) % G, (t - n_global) // G
        c, rr = flat["center"][idx], np.abs(flat["radius"][idx])
        fx, fz = (c[:, 0] - x0) * inv, (c[:, 2] - z0) * inv
        # (the host clamps a centre beyond the last cell into it; the kernel clamps the same way)
        assert np.all(fx >= ix - 1e-4) and np.all((fx <= ix + 1 + 1e-4) | (ix == G - 1))
        assert np.all(fz >= iz - 1e-4) and np.all((fz <= iz + 1 + 1e-4) | (iz == G - 1))
        assert np.all(rr <= pad) and np.all(c[:, 1] - rr >= ylo) and np.all(c[:, 1] + rr <= yhi)
        assert np.all(c[:, 0] <= x1) and np.all(c[:, 2] <= z1) and np.all(c[:, 0] >= x0) and np.all(c[:, 2] >= z0)


def ulp_jitter(x, rng):
    return (x * (f32(1.0) + rng.integers(-1, 2, x.shape).astype(f32) * f32(2.0 ** -23))).astype(f32)


def model_grid_cells(o, d, g, G, scale, rng, shrink=0.25):
    """rt_device.hpp grid_cells in numpy.float32; margins multiplied by `shrink`.
    -> (ix0, ix1, iz0, iz1, kind) with kind -1 cannot tell / 0 no cell / 1 a rectangle."""
    of, df = o.astype(f32), d.astype(f32)
    o1 = (np.abs(of[:, 0]) + np.abs(of[:, 1]) + np.abs(of[:, 2])).astype(f32)
    e = (f32(1e-6 * shrink) * (o1 + f32(scale))).astype(f32)
    dmin, dmax = np.abs(df).min(1), np.abs(df).max(1)
    sane = (dmin > f32(1e-30)) & (dmax < f32(1e15)) & (o1 < f32(1e15))
    m = (g[7] + e).astype(f32)
    lo = np.stack([g[0] - m, g[5] - e, g[1] - m], 1).astype(f32)
    hi = np.stack([g[3] + m, g[6] + e, g[4] + m], 1).astype(f32)
    with np.errstate(all="ignore"):
        inv = ulp_jitter((f32(1.0) / df.astype(np.float64)).astype(f32), rng)
        t0 = ((lo - of) * inv).astype(f32)
        t1 = ((hi - of) * inv).astype(f32)
        t_in = np.maximum(f32(0.0), np.minimum(t0, t1).max(1)).astype(f32)
        t_out = np.maximum(t0, t1).min(1).astype(f32)
        miss = t_out < t_in * f32(1.0 - 1e-4 * shrink)
        far = ~(t_out < f32(1e30))
        m2 = (g[7] + f32(4.0) * e).astype(f32)
        xa, xb = (t_in * df[:, 0] + of[:, 0]).astype(f32), (t_out * df[:, 0] + of[:, 0]).astype(f32)
        za, zb = (t_in * df[:, 2] + of[:, 2]).astype(f32), (t_out * df[:, 2] + of[:, 2]).astype(f32)
        eps = f32(1e-3 * shrink)
        fx0 = (((np.minimum(xa, xb) - m2) - g[0]) * g[2] - eps).astype(f32)
        fx1 = (((np.maximum(xa, xb) + m2) - g[0]) * g[2] + eps).astype(f32)
        fz0 = (((np.minimum(za, zb) - m2) - g[1]) * g[2] - eps).astype(f32)
        fz1 = (((np.maximum(za, zb) + m2) - g[1]) * g[2] + eps).astype(f32)
        bad = ~((fx0 <= fx1) & (fz0 <= fz1))
        cl = lambda v: np.clip(np.floor(np.nan_to_num(v, nan=0.0, posinf=1e9, neginf=-1e9)), 0, G - 1).astype(np.int64)
        ix0, ix1, iz0, iz1 = cl(fx0), cl(fx1), cl(fz0), cl(fz1)
        # the footprint row by row (GridSeg / grid_row_run: grids of more than 64 cells): columns of every grid row
        Xa, Xb = ((xa - g[0]) * g[2]).astype(f32), ((xb - g[0]) * g[2]).astype(f32)
        Za, Zb = ((za - g[1]) * g[2]).astype(f32), ((zb - g[1]) * g[2]).astype(f32)
        m = ((g[7] + f32(10.0) * e) * g[2] + f32(1e-3 * shrink)).astype(f32)
        dz = (Zb - Za).astype(f32)
        SL = ((Xb - Xa) * ulp_jitter((f32(1.0) / dz.astype(np.float64)).astype(f32), rng)).astype(f32)
        whole = ~((np.abs(dz) >= f32(1e-2)) & (np.abs(SL) < f32(1e6)))
        neg = ~whole & (SL < 0)
        za_, zb_ = np.where(neg, -Za, Za).astype(f32), np.where(neg, -Zb, Zb).astype(f32)
        sg = np.where(neg, f32(-1.0), f32(1.0)).astype(f32)
        alo = (np.where(neg, -(f32(1.0) + m), -m).astype(f32) - za_).astype(f32)
        ahi = (np.where(neg, m, f32(1.0) + m).astype(f32) - za_).astype(f32)
        dmin, dmax = np.minimum((zb_ - za_).astype(f32), f32(0.0)), np.maximum((zb_ - za_).astype(f32), f32(0.0))
        sl = np.where(whole, f32(0.0), np.abs(SL)).astype(f32)
        xl = (np.where(whole, np.minimum(Xa, Xb), (Xa - f32(1e-2 * shrink)).astype(f32)).astype(f32) - m).astype(f32)
        xh = (np.where(whole, np.maximum(Xa, Xb), (Xa + f32(1e-2 * shrink)).astype(f32)).astype(f32) + m).astype(f32)
        rows = np.arange(G, dtype=np.float32)[None, :]
        fma = lambda a, b, c: (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)   # (one rounding, like v_fma_f32)
        dl = np.maximum(fma(sg[:, None], rows, alo[:, None]), dmin[:, None])
        dh = np.minimum(fma(sg[:, None], rows, ahi[:, None]), dmax[:, None])
        lo, hi = fma(dl, sl[:, None], xl[:, None]), fma(dh, sl[:, None], xh[:, None])
        clc = lambda v: np.clip(np.floor(np.nan_to_num(v, nan=0.0, posinf=1e9, neginf=-1e9)).astype(np.int64), ix0[:, None], ix1[:, None])
        model_grid_cells.row_runs = (clc(lo), clc(hi))          # [rays, G] each; valid for rows iz0..iz1
    kind = np.where(~sane, -1, np.where(miss, 0, np.where(far | bad, -1, 1)))
    return ix0, ix1, iz0, iz1, kind


def reference_hits(o, d, c, r, t_min=0.001):
    """sphere.rs:16-34 in f64 for every (ray, sphere) pair -> bool [rays, spheres]: a root with t >= t_min exists."""
    oc = o[:, None, :] - c[None, :, :]
    a = (d * d).sum(1)[:, None]
    half_b = (oc * d[:, None, :]).sum(2)
    cc = (oc * oc).sum(2) - (r * r)[None, :]
    disc = half_b * half_b - a * cc
    with np.errstate(all="ignore"):
        sq = np.sqrt(np.where(disc >= 0, disc, 0.0))
        r1, r2 = (-half_b - sq) / a, (-half_b + sq) / a
    return (disc >= 0) & ((r1 >= t_min) | (r2 >= t_min))


def rays_for(flat, g, rng, n):
    """Origins on and around the scene's spheres, on the ground, at the book's camera; directions of every kind."""
    c, r = flat["center"], np.abs(flat["radius"])
    pick = rng.integers(0, len(flat), n)
    u = rng.normal(size=(n, 3))
    u /= np.linalg.norm(u, axis=1)[:, None]
    o = c[pick] + u * r[pick][:, None] * rng.choice([1.0, 1.0, 1.5, 4.0, 30.0], n)[:, None]      # on a surface, near, far
    ground = rng.random(n) < 0.3
    o[ground] = np.stack([rng.uniform(g[0] - 5, g[3] + 5, ground.sum()), np.zeros(ground.sum()), rng.uniform(g[1] - 5, g[4] + 5, ground.sum())], 1)
    cam = rng.random(n) < 0.2
    o[cam] = (13.0, 2.0, 3.0)
    d = rng.normal(size=(n, 3))
    grazing = rng.random(n) < 0.3
    d[grazing, 1] *= 0.01                                   # nearly horizontal: long footprints
    axis = rng.random(n) < 0.1
    d[axis] *= rng.choice([1.0, 1e-6, 1e-12], (axis.sum(), 3))                                   # nearly axis-parallel
    toward = rng.random(n) < 0.3
    tgt = rng.integers(0, len(flat), n)
    d[toward] = (c[tgt] + rng.normal(size=(n, 3)) * r[tgt][:, None] * 0.7 - o)[toward]          # aimed at some sphere
    d *= rng.choice([1.0, 1e-3, 1e3], n)[:, None]           # the reference never normalises its directions
    return o, d


@pytest.mark.parametrize("name", ["book", "tenk", "cloud", "clusters", "line"])
def test_a_sphere_the_reference_can_hit_lies_in_a_cell_of_the_rays_footprint(name):
    flat = CASES[name]
    G, n_global, g, slot_of = layout(flat)
    assert G > 0
    rng = np.random.default_rng(5)
    # the kernel's scale is at least the grid box's own (rt_api.hip); use exactly that: the smallest margins
    scale = max(abs(g[0]), abs(g[3])) + max(abs(g[5]), abs(g[6])) + max(abs(g[1]), abs(g[4])) + 2 * g[7]
    cell_slots = np.arange(32 * n_global, len(slot_of))
    cell_slots = cell_slots[slot_of[cell_slots] >= 0]
    idx = slot_of[cell_slots]
    cell = cell_slots // 32 - n_global
    six, siz = cell % G, cell // G
    c, r = flat["center"][idx], np.abs(flat["radius"][idx])
    n_rays, checked, rect_cells, line_cells = 6000, 0, [], []
    for lo in range(0, n_rays, 500):
        o, d = rays_for(flat, g, rng, 500)
        ix0, ix1, iz0, iz1, kind = model_grid_cells(o, d, g, G, scale, rng)
        hits = reference_hits(o, d, c, r)
        inside = (six[None, :] >= ix0[:, None]) & (six[None, :] <= ix1[:, None]) & (siz[None, :] >= iz0[:, None]) & (siz[None, :] <= iz1[:, None])
        n_cells = (ix1 - ix0 + 1) * (iz1 - iz0 + 1)
        if n_global + G * G > 64:
            # the kernel instantiation for large grids marks the cells row by row: the sphere's column must lie in the run
            # of the sphere's row (and the rows are those of the rectangle)
            rlo, rhi = model_grid_cells.row_runs
            ray = np.arange(len(o))[:, None]
            inside &= (six[None, :] >= rlo[ray, siz[None, :]]) & (six[None, :] <= rhi[ray, siz[None, :]])
            rowsel = (np.arange(G)[None, :] >= iz0[:, None]) & (np.arange(G)[None, :] <= iz1[:, None])
            n_row_cells = np.where(rowsel, rhi - rlo + 1, 0).sum(1)
            assert np.all(n_row_cells[kind == 1] <= n_cells[kind == 1])         # never more than the rectangle
            line_cells.append(np.where(kind == 1, n_row_cells, 0))
            n_cells = n_row_cells
        ok = (kind[:, None] == -1) | ((kind[:, None] == 1) & inside)
        bad = hits & ~ok
        assert not bad.any(), (name, np.argwhere(bad)[:5], o[np.argwhere(bad)[0][0]], d[np.argwhere(bad)[0][0]])
        checked += int(hits.sum())
        rect_cells.append(np.where(kind == 1, n_cells, 0))
    assert checked > 2000                                    # the statement was tested on real hits
    # and the footprints are not trivially "everything": the typical ray marks a small part of the grid
    rc = np.concatenate(rect_cells)
    assert np.median(rc[rc > 0]) <= max(4, G * G // 4)
    if line_cells:
        print(name, "cells per footprint: mean", np.concatenate(line_cells)[rc > 0].mean())


def test_the_model_can_fail():
    """With the pad removed a sphere straddling a cell border is missed: the check above has teeth."""
    flat = CASES["book"]
    G, n_global, g, slot_of = layout(flat)
    rng = np.random.default_rng(5)
    g0 = g.copy()
    g0[7] = -0.15                                            # shrink instead of grow
    cell_slots = np.arange(32 * n_global, len(slot_of))
    cell_slots = cell_slots[slot_of[cell_slots] >= 0]
    idx = slot_of[cell_slots]
    cell = cell_slots // 32 - n_global
    six, siz = cell % G, cell // G
    c, r = flat["center"][idx], np.abs(flat["radius"][idx])
    o, d = rays_for(flat, g, rng, 3000)
    ix0, ix1, iz0, iz1, kind = model_grid_cells(o, d, g0, G, 40.0, rng)
    hits = reference_hits(o, d, c, r)
    inside = (six[None, :] >= ix0[:, None]) & (six[None, :] <= ix1[:, None]) & (siz[None, :] >= iz0[:, None]) & (siz[None, :] <= iz1[:, None])
    ok = (kind[:, None] == -1) | ((kind[:, None] == 1) & inside)
    assert (hits & ~ok).any()
