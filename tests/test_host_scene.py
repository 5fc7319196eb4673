"""Host mirror of the reference's scene interface (rtiow_amd.scene) and of the
row sharding, checked against the oracle's own restatements and the fixtures."""
import os

import numpy as np
import pytest

import rtiow_amd as rt
from rtiow_amd.distributed import shard_row_map
from conftest import GOLDEN


def test_random_scene_matches_committed_fixture(book1_flat):
    flat = rt.random_scene(1).flatten()
    assert flat.tobytes() == book1_flat.tobytes()


def test_random_scene_structure():                  # main.rs:59-102
    world = rt.random_scene(1)
    flat = world.flatten()
    assert len(flat) <= 1 + 23 * 23 + 3 and len(flat) >= 520
    g = flat[0]
    assert g["center"].tolist() == [0, -1000, 0] and g["radius"] == 1000 and g["kind"] == 0
    assert flat[-3]["kind"] == 2 and flat[-3]["param"] == 1.5 and flat[-3]["center"].tolist() == [0, 1, 0]
    assert flat[-2]["kind"] == 0 and flat[-2]["albedo"].tolist() == [0.4, 0.2, 0.1]
    assert flat[-1]["kind"] == 1 and flat[-1]["param"] == 0.0 and flat[-1]["center"].tolist() == [4, 1, 0]
    small = flat[1:-3]
    assert (small["radius"] == 0.2).all() and (small["center"][:, 1] == 0.2).all()
    d = small["center"] - np.array([4, 0.2, 0])
    assert (np.sqrt((d * d).sum(1)) > 0.9).all()    # main.rs:72
    mix = np.bincount(small["kind"], minlength=3) / len(small)
    assert 0.72 < mix[0] < 0.88 and 0.08 < mix[1] < 0.22 and 0.01 < mix[2] < 0.10
    metal = small[small["kind"] == 1]
    assert (metal["albedo"] >= 0.5).all() and (metal["albedo"] < 1.0).all()
    assert (metal["param"] >= 0.0).all() and (metal["param"] < 0.5).all()
    assert (small[small["kind"] == 2]["param"] == 1.5).all()


def test_other_seeds_give_other_scenes():
    assert rt.random_scene(1).flatten().tobytes() != rt.random_scene(2).flatten().tobytes()


def test_tenk_scene_matches_fixture_head():
    fx = np.load(os.path.join(GOLDEN, "tenk_scene_seed1_head.npz"), allow_pickle=False)
    flat = rt.random_scene(1, grid=(-50, 49)).flatten()
    assert len(flat) == int(fx["count"]) and len(flat) > 9900
    assert flat[:8].tobytes() == fx["head"].tobytes() and flat[-8:].tobytes() == fx["tail"].tobytes()


@pytest.mark.parametrize("wh", [(400, 225), (1200, 800), (3840, 2160)])
def test_camera_new_equals_oracle_restatement(oracle_mod, wh):   # camera.rs:17-45
    w, h = wh
    host = rt.book1_camera(w, h)
    orc = oracle_mod.book1_camera(w, h)
    for name in ("origin", "lower_left_corner", "horizontal", "vertical", "u", "v"):
        assert list(getattr(orc, name)) == [float(x) for x in getattr(host, name)], name
    assert orc.lens_radius == host.lens_radius == 0.05
    c = host.to_rt_camera()
    assert list(c.horizontal) == list(orc.horizontal)


def test_hittable_list_keeps_push_order():
    w = rt.HittableList()
    w.push(rt.Sphere(rt.Point3(1, 2, 3), 0.5, rt.Metal(rt.Color(0.1, 0.2, 0.3), 0.25)))
    w.push(rt.Sphere(rt.Point3(4, 5, 6), 1.5, rt.Dialectric(1.5)))
    f = w.flatten()
    assert f["center"].tolist() == [[1, 2, 3], [4, 5, 6]] and f["kind"].tolist() == [1, 2]
    assert f["param"].tolist() == [0.25, 1.5] and f.dtype.itemsize == 72


@pytest.mark.parametrize("height,tile,count", [(675, 8, 8), (225, 8, 3), (18, 4, 2), (7, 3, 4), (5, 8, 2), (2160, 16, 8)])
def test_shard_rows_partition_the_image(height, tile, count):
    seen = []
    for k in range(count):
        rows = shard_row_map(height, tile, k, count)
        p = rt.make_params(64, height, 1, tile_rows=tile, shard_index=k, shard_count=count)
        assert rt.shard_rows(p) == len(rows)                        # C ABI (no GPU needed) agrees
        assert rt.shard_row_indices(p).tolist() == rows.tolist()
        assert (np.diff(rows) > 0).all() if len(rows) > 1 else True
        seen.extend(rows.tolist())
    assert sorted(seen) == list(range(height))


def test_ppm_roundtrip(tmp_path):
    rgba = (np.arange(5 * 7 * 4) % 251).astype(np.uint8).reshape(5, 7, 4)
    path = tmp_path / "x.ppm"
    rt.write_ppm(str(path), rgba)
    assert np.array_equal(rt.read_ppm(str(path)), rgba[:, :, :3])


def test_scene_file_roundtrip(tmp_path, book1_flat):
    path = str(tmp_path / "scene.bin")
    rt.save_scene(path, rt.random_scene(1))
    got = rt.load_scene(path)
    assert got.tobytes() == book1_flat.tobytes()
    with open(path, "ab") as f:
        f.write(b"\0" * 5)
    with pytest.raises(ValueError):
        rt.load_scene(path)


def test_checkpoint_roundtrip(tmp_path):
    fix = (np.arange(4 * 5 * 3, dtype=np.uint64) * np.uint64(1 << 33)).reshape(4, 5, 3)
    path = str(tmp_path / "ck.npz")
    rt.save_checkpoint(path, fix, 37, 0xDEADBEEF12345678)
    f2, spp, seed = rt.load_checkpoint(path)
    assert np.array_equal(f2, fix) and spp == 37 and seed == 0xDEADBEEF12345678
