"""The C++ host mirror (host/rtiow_host.hpp + host/rtiow_render.cpp): same scene bytes as the
Python mirror (CPU), and the same image through the C ABI (GPU)."""
import os
import subprocess

import numpy as np
import pytest

import rtiow_amd as rt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "host", "rtiow_render")


def _cli():
    if not os.path.exists(CLI):
        import __graft_entry__ as g
        g.build_host_cli()
    return CLI


@pytest.mark.parametrize("args,grid", [([], (-11, 11)), (["--grid", "-50", "49"], (-50, 49)), (["--scene-seed", "7"], (-11, 11))])
def test_cpp_random_scene_equals_python(tmp_path, args, grid):
    path = str(tmp_path / "scene.bin")
    subprocess.run([_cli(), "--dump-scene", path, *args], check=True, capture_output=True)
    got = np.fromfile(path, dtype=rt.SPHERE_DTYPE)
    seed = 7 if "--scene-seed" in args else 1
    want = rt.random_scene(seed, grid=grid).flatten()
    assert got.tobytes() == want.tobytes()


@pytest.mark.parametrize("w,h", [(301, 207), (1, 1), (5, 40000)])
def test_cpp_png_writer_on_a_fixed_pattern(tmp_path, w, h):
    """host/rtiow_host.hpp write_png (no zlib: stored DEFLATE blocks, own CRC-32 and Adler-32): `--test-png` sends a fixed pattern through
    it -- more than one 65 535-byte block, a single pixel, a tall image -- and rtiow_amd.read_png (zlib, CRCs checked) and PIL must get
    the pattern back."""
    out = str(tmp_path / "p.png")
    subprocess.run([_cli(), "--test-png", str(w), str(h), out], check=True, capture_output=True)
    y, x = np.mgrid[0:h, 0:w]
    want = np.stack([(7 * x + 13 * y) & 255, (x ^ y) & 255, (x * y) & 255, np.full_like(x, 255)], -1).astype(np.uint8)
    assert np.array_equal(rt.read_png(out), want)
    try:
        from PIL import Image
    except ImportError:
        return
    assert np.array_equal(np.array(Image.open(out)), want)


@pytest.mark.gpu
def test_cpp_cli_saves_the_references_image_png(tmp_path, renderer, book1_flat):
    """main.rs:177: `--out image.png` holds the RGBA8 frame, rows top first, alpha 255 -- byte for byte the library's resolve."""
    out = str(tmp_path / "image.png")
    subprocess.run([_cli(), "--width", "160", "--height", "90", "--spp", "6", "--out", out], check=True, capture_output=True)
    renderer.upload_scene(book1_flat)
    _, fix, _ = renderer.render(rt.book1_camera(160, 90), rt.make_params(160, 90, 6))
    want = renderer.resolve_rgba8(fix, 6, flip=True)
    got = rt.read_png(out)
    assert got.shape == (90, 160, 4) and (got[:, :, 3] == 255).all() and np.array_equal(got, want)


@pytest.mark.gpu
def test_cpp_cli_renders_the_same_image(tmp_path, renderer, book1_flat):
    out = str(tmp_path / "image.ppm")
    subprocess.run([_cli(), "--width", "160", "--height", "90", "--spp", "6", "--out", out], check=True, capture_output=True)
    renderer.upload_scene(book1_flat)
    _, fix, _ = renderer.render(rt.book1_camera(160, 90), rt.make_params(160, 90, 6))
    want = renderer.resolve_rgba8(fix, 6, flip=True)[:, :, :3]
    assert np.array_equal(rt.read_ppm(out), want)


@pytest.mark.gpu
def test_cpp_cli_reads_a_scene_file(tmp_path, renderer):
    world = rt.random_scene(5)
    path = str(tmp_path / "s.bin")
    rt.save_scene(path, world)
    out = str(tmp_path / "image.ppm")
    subprocess.run([_cli(), "--scene", path, "--width", "96", "--height", "54", "--spp", "3", "--out", out],
                   check=True, capture_output=True)
    renderer.upload_scene(world)
    _, fix, _ = renderer.render(rt.book1_camera(96, 54), rt.make_params(96, 54, 3))
    assert np.array_equal(rt.read_ppm(out), renderer.resolve_rgba8(fix, 3, flip=True)[:, :, :3])


@pytest.mark.gpu
@pytest.mark.parametrize("extra,copies", [(["--devices", "0,0"], 2), (["--devices", "0,0,0", "--tile-rows", "7"], 4),
                                          (["--devices", "0", "--force-rccl"], 1), (["--devices", "0,0,0,0,0", "--tile-rows", "40"], 5)])
def test_cpp_sharded_render_equals_the_single_context_image(tmp_path, extra, copies):
    """host/rtiow_multi.hpp: one rt_context per listed device, each on its own host thread (here: two or three
    contexts on device 0 rendering their shards CONCURRENTLY -- the threading rule of include/rtiow_hip.h:
    distinct contexts may be used from distinct threads at the same time), rows gathered and put back in
    image order -- ONE 2D device copy per shard plus one for a ragged last tile (169 rows: 24 tiles of 7 + 1 row; 4 tiles
    of 40 + 9 rows, so the fifth shard owns the ragged tile only), not one per tile -- one resolve.  The result must be
    the single-context image byte for byte.  A list of distinct
    devices gathers with RCCL ncclGather; with one GPU here that path runs as a communicator of one
    (--force-rccl)."""
    single, sharded = str(tmp_path / "a.ppm"), str(tmp_path / "b.ppm")
    size = ["--width", "300", "--height", "169", "--spp", "40"]
    subprocess.run([_cli(), *size, "--out", single], check=True, capture_output=True)
    r = subprocess.run([_cli(), *size, *extra, "--out", sharded], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert open(single, "rb").read() == open(sharded, "rb").read()
    assert f": {copies} device copies" in r.stdout, r.stdout


@pytest.mark.parametrize("H,T,n", [(169, 7, 3), (169, 1, 2), (169, 40, 5), (4320, 1, 8), (4320, 16, 8), (675, 8, 1), (5, 8, 3), (10, 3, 4)])
def test_cpp_reassembly_plan_is_the_inverse_of_the_shard_row_map(H, T, n):
    """host/rtiow_multi.hpp reassembly_plan(): the strided copies rank 0 issues must send compact row c of shard k to
    image row rt_shard_row_index(c) -- every image row exactly once -- with at most two copies per shard (one 2D copy for
    the shard's full tiles, one for a ragged last tile): 8 calls for 7680x4320 on 8 GPUs at tiles of one row, not 4 320."""
    from rtiow_amd import render as rr
    from rtiow_amd.distributed import shard_row_map
    out = subprocess.run([_cli(), "--reassembly-plan", str(H), str(T), str(n)], check=True, capture_output=True, text=True).stdout
    plan = [tuple(int(x) for x in line.split()) for line in out.splitlines()]
    assert len(plan) <= n + 1 and all(sum(1 for e in plan if e[0] == k) <= 2 for k in range(n))
    if (H, T, n) == (4320, 1, 8):
        assert len(plan) == 8
    owner = -np.ones(H, dtype=np.int64)
    for k in range(n):
        # image row of every compact row of shard k: the library's own map (rt_shard_row_index) == the Python gatherer's
        rows = rr.shard_row_indices(rr.make_params(16, H, 1, tile_rows=T, shard_index=k, shard_count=n))
        assert np.array_equal(rows, shard_row_map(H, T, k, n))
        got = -np.ones(len(rows), dtype=np.int64)
        for (kk, dst, src, cnt, pieces, dp, sp) in plan:
            if kk != k:
                continue
            for i in range(pieces):
                for r in range(cnt):
                    assert got[src + i * sp + r] == -1
                    got[src + i * sp + r] = dst + i * dp + r
        assert np.array_equal(got, np.asarray(rows))
        assert np.all(owner[np.asarray(rows, dtype=np.int64)] == -1)
        owner[np.asarray(rows, dtype=np.int64)] = k
    assert np.all(owner >= 0)


@pytest.mark.parametrize("devices", ["0,99", "99", "0,0,99"])
def test_cpp_sharded_render_with_a_bad_device_reports_instead_of_blocking(tmp_path, devices):
    """host/rtiow_multi.hpp: a shard that cannot start (no such device) makes render_sharded return an error -- every
    shard finishes its launch phase before any enters the gather, so no rank is left waiting in a collective, and the
    single exit path frees what the others had created.  Runs with or without a GPU (without one every shard fails)."""
    r = subprocess.run([_cli(), "--width", "64", "--height", "36", "--spp", "2", "--devices", devices, "--out", str(tmp_path / "x.ppm")],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "render_sharded failed" in r.stderr
    assert not os.path.exists(str(tmp_path / "x.ppm"))


@pytest.mark.gpu
def test_cpp_sharded_render_with_a_rejected_scene_reports_instead_of_blocking(tmp_path):
    """The same on the RCCL path with a live communicator: rt_upload_scene rejects the scene (unknown material kind) in
    the launch phase of every shard; the process must return the error, not hang in ncclGather."""
    bad = rt.random_scene(1).flatten()
    bad["kind"][5] = 9
    path = str(tmp_path / "bad.bin")
    bad.tofile(path)
    r = subprocess.run([_cli(), "--scene", path, "--width", "64", "--height", "36", "--spp", "2", "--devices", "0", "--force-rccl",
                        "--out", str(tmp_path / "x.ppm")], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "unknown material kind" in r.stderr
