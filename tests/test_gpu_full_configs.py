"""Every BASELINE.json configuration at its TRUE size and spp, against the oracle.

The oracle cannot render a 4-G-sample frame in seconds, but it can render single image rows of
it: with tile_rows = 1, shard_count = H, shard_index = j the GPU renders image row j alone (same
global pixel keys, same samples as in the full frame: main.rs:122-139 row by row), and Oracle B
renders the same row at full spp.  Bar: bit-exact (u64 sums and ray counts).  The full frames are
then tied to those rows: the probed rows of a full-frame (or full-shard) launch must equal the
row-probe results, the frame must be reproducible, and an 8-way row-sharded render must
reassemble to it bit for bit.

  configs[2]  book-1 final scene, 3840x2160, 500 spp
  configs[3]  10k random spheres, 1920x1080, 256 spp
  configs[4]  book-1 final scene, 7680x4320, 1000 spp: rows, and ONE of its 8 shards (what one GPU
              of the 8-GPU run renders) in full
"""
import numpy as np
import pytest

import rtiow_amd as rt
from rtiow_amd.distributed import shard_row_map

pytestmark = pytest.mark.gpu


def probe_row(renderer, oracle_mod, flat, w, h, spp, j):
    """(GPU fix [W,3], oracle fix [W,3], GPU rays, oracle rays) of image row j at full size and spp."""
    cam = rt.book1_camera(w, h)
    p = rt.make_params(w, h, spp, seed=1, tile_rows=1, shard_index=j, shard_count=h)
    assert rt.shard_rows(p) == 1 and list(rt.shard_row_indices(p)) == [j]
    _, fix, st = renderer.render(cam, p)
    op = oracle_mod.make_params(w, h, spp, seed=1, rows=(j, j + 1, 1))
    fb, _, stb = oracle_mod.render_b(oracle_mod.camera_from_host(cam), flat, op)
    assert st["samples"] == w * spp
    return fix[0], fb[0], st["rays_traced"], stb["rays_traced"]


@pytest.fixture(scope="module")
def cfg3_rows(renderer, oracle_mod, book1_flat):
    renderer.upload_scene(book1_flat)
    out = {}
    for j in (0, 700, 1080, 2159):
        g, o, rg, ro = probe_row(renderer, oracle_mod, book1_flat, 3840, 2160, 500, j)
        out[j] = (g, o, rg, ro)
    return out


def test_cfg3_rows_bit_exact_vs_oracle(cfg3_rows):
    """configs[2] (3840x2160x500): bottom row, a ground row through the small spheres, a row through
    the three big spheres, the top (sky) row."""
    for j, (g, o, rg, ro) in cfg3_rows.items():
        assert np.array_equal(g, o), f"row {j}"
        assert rg == ro, f"row {j}"
    assert cfg3_rows[2159][2] == 3840 * 500                       # top row: sky only, one ray per sample


def test_cfg3_full_frame_rows_determinism_and_8way_reassembly(renderer, book1_flat, cfg3_rows):
    w, h, spp = 3840, 2160, 500
    renderer.upload_scene(book1_flat)
    cam = rt.book1_camera(w, h)
    _, full, st = renderer.render(cam, rt.make_params(w, h, spp))
    assert st["samples"] == w * h * spp
    for j, (g, _, _, _) in cfg3_rows.items():                     # the frame holds the oracle-checked rows
        assert np.array_equal(full[j], g), f"row {j}"
    _, again, st2 = renderer.render(cam, rt.make_params(w, h, spp))
    assert np.array_equal(again, full) and st2["rays_traced"] == st["rays_traced"]
    del again
    asm = np.zeros_like(full)
    rays = 0
    for k in range(8):                                            # the 8 ranks' shards, one after the other
        p = rt.make_params(w, h, spp, tile_rows=1, shard_index=k, shard_count=8)
        _, part, stk = renderer.render(cam, p)
        asm[shard_row_map(h, 1, k, 8)] = part
        rays += stk["rays_traced"]
    assert np.array_equal(asm, full) and rays == st["rays_traced"]


def test_cfg4_rows_and_full_frame(renderer, oracle_mod):
    """configs[3]: 10 001 spheres, 1920x1080 at the full 256 spp."""
    w, h, spp = 1920, 1080, 256
    flat = rt.random_scene(1, grid=(-50, 49)).flatten()
    assert len(flat) == 10001
    renderer.upload_scene(flat)
    rows = {}
    for j in (330, 760):
        g, o, rg, ro = probe_row(renderer, oracle_mod, flat, w, h, spp, j)
        assert np.array_equal(g, o), f"row {j}"
        assert rg == ro
        rows[j] = g
    cam = rt.book1_camera(w, h)
    _, full, st = renderer.render(cam, rt.make_params(w, h, spp))
    assert st["samples"] == w * h * spp and st["n_spheres"] == 10001
    for j, g in rows.items():
        assert np.array_equal(full[j], g), f"row {j}"
    _, again, st2 = renderer.render(cam, rt.make_params(w, h, spp))
    assert np.array_equal(again, full) and st2["rays_traced"] == st["rays_traced"]


def test_cfg5_rows_and_one_of_the_eight_shards(renderer, oracle_mod, book1_flat):
    """configs[4] (7680x4320x1000, row-tiled over 8 GPUs): two rows against the oracle, then the
    whole shard rank 3 of 8 renders (540 rows, 4.147 G samples), which must hold its probed row and be
    reproducible."""
    w, h, spp = 7680, 4320, 1000
    renderer.upload_scene(book1_flat)
    rows = {}
    for j in (1403, 3000):                                        # 1403 = 8*175 + 3: a row of shard 3
        g, o, rg, ro = probe_row(renderer, oracle_mod, book1_flat, w, h, spp, j)
        assert np.array_equal(g, o), f"row {j}"
        assert rg == ro
        rows[j] = g
    cam = rt.book1_camera(w, h)
    p = rt.make_params(w, h, spp, tile_rows=1, shard_index=3, shard_count=8)
    owned = shard_row_map(h, 1, 3, 8)
    assert len(owned) == 540 and owned[175] == 1403
    _, part, st = renderer.render(cam, p)
    assert part.shape == (540, w, 3) and st["samples"] == 540 * w * spp
    assert np.array_equal(part[175], rows[1403])
    _, again, st2 = renderer.render(cam, p)
    assert np.array_equal(again, part) and st2["rays_traced"] == st["rays_traced"]
