"""Size-independent properties at BASELINE.json's full sizes (the oracle cannot
finish these in seconds): determinism, sharding invariance, additive passes,
work-decomposition invariance, sample accounting -- and the reference's own PNG
sky rows reproduced by the GPU."""
import json
import os

import numpy as np
import pytest

import rtiow_amd as rt
from rtiow_amd.distributed import shard_row_map
from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cfg2(renderer, book1_flat):
    """BASELINE.json configs[1]: book-1 final scene, 1200x675, 100 spp, depth 50."""
    renderer.upload_scene(book1_flat)
    w, h, spp = 1200, 675, 100
    sm, fix, st = renderer.render(rt.book1_camera(w, h), rt.make_params(w, h, spp))
    return w, h, spp, fix, st


def test_cfg2_accounting(cfg2, book1_flat):
    w, h, spp, fix, st = cfg2
    assert st["samples"] == w * h * spp
    assert 2.4 < st["rays_traced"] / st["samples"] < 2.9
    assert st["sphere_tests"] == st["rays_traced"] * len(book1_flat)
    assert st["scan_mode"] == int(os.environ.get("RTIOW_SCAN_MODE", "5"))     # the tube filter unless overridden
    mean = fix.astype(np.float64) / 2.0 ** 32 / spp
    assert 0.0 <= mean.min() and mean.max() <= 1.0 + 1e-12          # sky <= 1 and albedos <= 1
    assert 0.3 < mean.mean() < 0.6
    # A wave keeps the sums of at most 2 unfinished work blocks in LDS (4 on the general kernel); a path of more than ~15 (~30) bounces
    # holds its block open for longer than that, and the block's last samples then go to the frame buffer one by one.
    # They exist at this size (6153 of these 81 M paths run the full 50 bounces; 1.4 in 1 000 samples go this way) and they are rare.
    assert 0 < st["direct_samples"] < st["samples"] // 400


@pytest.mark.parametrize("max_depth", [50, 5, 1])
def test_rays_per_bounce_histogram(renderer, book1_flat, max_depth):
    """rt_stats.live_per_bounce (RT_FLAG_DIAG_STATS): one camera ray per sample, never more rays at a deeper
    bounce, nothing at or beyond max_depth, and the histogram sums to rays_traced."""
    renderer.upload_scene(book1_flat)
    w, h, spp = 200, 112, 8
    _, _, st = renderer.render(rt.book1_camera(w, h), rt.make_params(w, h, spp, max_depth=max_depth, flags=rt.RT_FLAG_DIAG_STATS))
    live = st["live_per_bounce"]
    assert len(live) == 64 and sum(live) == st["rays_traced"]
    assert live[0] == st["samples"] == w * h * spp
    assert all(a >= b for a, b in zip(live, live[1:]))
    assert all(v == 0 for v in live[min(max_depth, 63):]) if max_depth < 64 else True
    if max_depth >= 5:
        assert 0 < live[4] < live[1] < live[0]
    _, _, st0 = renderer.render(rt.book1_camera(w, h), rt.make_params(w, h, spp, max_depth=max_depth))
    assert sum(st0["live_per_bounce"]) == 0 and st0["rays_traced"] == st["rays_traced"]       # off without the flag


def test_cfg2_is_deterministic(renderer, book1_flat, cfg2):
    w, h, spp, fix, st = cfg2
    renderer.upload_scene(book1_flat)
    _, again, st2 = renderer.render(rt.book1_camera(w, h), rt.make_params(w, h, spp))
    assert np.array_equal(again, fix) and st2["rays_traced"] == st["rays_traced"]


@pytest.mark.parametrize("count,tile", [(8, 8), (3, 16), (2, 1)])
def test_cfg2_row_shards_reassemble_bit_identically(renderer, book1_flat, cfg2, count, tile):
    """The 8-way row-tiled result must equal the 1-GPU result:
    rendered here as `count` shards one after the other on one GPU."""
    w, h, spp, fix, st = cfg2
    renderer.upload_scene(book1_flat)
    full = np.zeros_like(fix)
    rays = 0
    for k in range(count):
        p = rt.make_params(w, h, spp, tile_rows=tile, shard_index=k, shard_count=count)
        _, part, stk = renderer.render(rt.book1_camera(w, h), p)
        rows = shard_row_map(h, tile, k, count)
        assert part.shape[0] == len(rows)
        full[rows] = part
        rays += stk["rays_traced"]
    assert np.array_equal(full, fix) and rays == st["rays_traced"]


def test_cfg2_passes_are_additive(renderer, book1_flat, cfg2):
    """spp split into passes with sample_begin; exact integer sums make it bit-identical."""
    w, h, spp, fix, st = cfg2
    renderer.upload_scene(book1_flat)
    cam = rt.book1_camera(w, h)
    acc = np.zeros_like(fix)
    for begin, n in ((0, 37), (37, 1), (38, 62)):
        _, part, _ = renderer.render(cam, rt.make_params(w, h, n, sample_begin=begin))
        acc += part
    assert np.array_equal(acc, fix)


def test_cfg2_device_buffers_accumulate_flag(renderer, book1_flat, cfg2):
    torch = pytest.importorskip("torch")
    w, h, spp, fix, st = cfg2
    renderer.upload_scene(book1_flat)
    cam = rt.book1_camera(w, h)
    d_fix = torch.zeros((h, w, 3), dtype=torch.int64, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    renderer.render_device(cam, rt.make_params(w, h, 60), d_fix.data_ptr(), stream)
    renderer.render_device(cam, rt.make_params(w, h, 40, sample_begin=60, flags=rt.RT_FLAG_ACCUMULATE),
                           d_fix.data_ptr(), stream)
    d_sum = torch.empty((h, w, 3), dtype=torch.float32, device="cuda:0")
    renderer.fix_to_f32_device(d_fix.data_ptr(), d_fix.numel(), d_sum.data_ptr(), stream)
    d_rgba = torch.empty((h, w, 4), dtype=torch.uint8, device="cuda:0")
    renderer.resolve_rgba8_device(d_fix.data_ptr(), w, h, spp, 1, d_rgba.data_ptr(), stream)
    torch.cuda.synchronize()
    assert np.array_equal(d_fix.cpu().numpy().view(np.uint64), fix)
    assert np.array_equal(d_rgba.cpu().numpy(), renderer.resolve_rgba8(fix, spp, flip=True))
    want_sum = (fix.astype(np.float64) / 2.0 ** 32).astype(np.float32)
    assert np.array_equal(d_sum.cpu().numpy(), want_sum)


def test_blocks_per_cu_does_not_change_the_full_frame(book1_flat, cfg2):
    """Blocks per CU is a scheduling knob only (the knobs that select other code paths -- grid resolution, no grid, LDS
    block sums off -- are compared with the ORACLE in test_gpu_oracle_holes.py; here the full configs[1] frame)."""
    w, h, spp, fix, st = cfg2
    for env in ({"RTIOW_BLOCKS_PER_CU": "1"}, {"RTIOW_BLOCKS_PER_CU": "3", "RTIOW_RING_MIN_SPP": "101"}):
        os.environ.update(env)
        try:
            r = rt.Renderer(0)
            r.upload_scene(book1_flat)
            _, got, st2 = r.render(rt.book1_camera(w, h), rt.make_params(w, h, spp))
            r.close()
        finally:
            for k in env:
                os.environ.pop(k)
        assert np.array_equal(got, fix), env
        assert st2["rays_traced"] == st["rays_traced"]
        assert st2["grid_blocks"] == 256 * int(env["RTIOW_BLOCKS_PER_CU"])


def test_gpu_reproduces_the_reference_png_sky_rows(renderer, book1_flat):
    """The GPU render at the reference PNG's geometry (1200x800, 3:2) gives the PNG's own
    sky bytes: Camera, get_ray direction, sky gradient, to_rgba and the flip, end to end."""
    fx = json.load(open(os.path.join(GOLDEN, "ref_png_sky_rows.json")))
    w, h, spp = 1200, 800, 32
    renderer.upload_scene(book1_flat)
    cam = rt.Camera(rt.Point3(13, 2, 3), rt.Point3(0, 0, 0), rt.Vec3(0, 1, 0), 20.0, 3.0 / 2.0, 0.1, 10.0)
    _, fix, _ = renderer.render(cam, rt.make_params(w, h, spp))
    rgba = renderer.resolve_rgba8(fix, spp, flip=True)              # top row first, like the PNG
    assert (rgba[..., 3] == 255).all()
    for y in (0, 1, 4, 8, 24, 37, 50):
        assert (rgba[y, :, :3] == np.array(fx["constant_rows"][str(y)], dtype=np.uint8)).all(), y
    for pt in fx["points"]:
        if pt["x"] != 600:
            assert rgba[pt["y"], pt["x"], :3].tolist() == pt["rgb"]
    m = rgba[..., :3].reshape(-1, 3).mean(0)                        # loose sanity band only
    assert np.abs(m - np.array(fx["image_mean_rgb"])).max() < 15


def test_gpu_reproduces_what_the_reference_png_holds_about_the_big_spheres(renderer, book1_flat):
    """tests/png_pins.py on the GPU render (1200x800, 3:2, 48 spp): the sky mirrored in the fuzz-0 Metal sphere
    within 1 of the PNG, the silhouettes of the three hard-coded spheres and the ground's horizon where the PNG
    has them, the mean colour of the Lambertian sphere's sky-facing patch -- the reference's own artefact
    pinning Sphere::hit, HitRecord::new, Metal::scatter/reflect, Lambertian::scatter and -- through the refracted sky
    in the lower half of the glass sphere -- Dialectric::scatter on the device."""
    import png_pins
    fx = png_pins.fixture()
    w, h, spp = 1200, 800, 48
    cam = rt.Camera(rt.Point3(13, 2, 3), rt.Point3(0, 0, 0), rt.Vec3(0, 1, 0), 20.0, 3.0 / 2.0, 0.1, 10.0)
    renderer.upload_scene(book1_flat)
    _, fix, _ = renderer.render(cam, rt.make_params(w, h, spp))
    rgb = renderer.resolve_rgba8(fix, spp, flip=True)[..., :3]
    renderer.upload_scene(book1_flat[:0])                          # the empty scene: the sky reference
    _, fix0, _ = renderer.render(cam, rt.make_params(w, h, 4))
    sky = renderer.resolve_rgba8(fix0, 4, flip=True)[..., :3]
    y0, y1 = fx["rows_scanned"]
    png_pins.check_metal_cap(fx, lambda y: rgb[y])
    png_pins.check_silhouettes_and_horizon(fx, png_pins.nonsky_mask(fx, rgb[y0:y1], sky[y0:y1]))
    png_pins.check_lambertian_patch(fx, rgb[y0:y1])
    png_pins.check_dialectric_patch(fx, lambda a, b: rgb[a:b])


def test_all_scan_filters_give_the_same_bits(book1_flat, cfg2):
    """The filter implementations of the loaded library (product: VALU + scalar loads and the bf16x2 tube;
    cross-check build, see test_gpu_crosscheck_modes.py: also f32 MFMA, bf16x3 MFMA, single-contraction
    bf16x3 MFMA) are ways of discarding spheres the reference cannot hit: the frame must not depend on
    which runs."""
    from rtiow_amd import _ffi
    w, h, spp, fix, st = cfg2
    cands, roots = {}, set()
    modes = ("1", "2", "3", "4", "5") if _ffi.has_crosscheck_modes() else ("1", "5")
    for mode in modes:
        os.environ["RTIOW_SCAN_MODE"] = mode
        try:
            r = rt.Renderer(0)
            r.upload_scene(book1_flat)
            _, got, st2 = r.render(rt.book1_camera(w, h), rt.make_params(w, h, spp, flags=rt.RT_FLAG_DIAG_STATS))
            r.close()
        finally:
            os.environ.pop("RTIOW_SCAN_MODE")
        assert np.array_equal(got, fix), mode
        assert st2["rays_traced"] == st["rays_traced"] and st2["scan_mode"] == int(mode)
        roots.add(st2["exact_roots"])
        cands[mode] = st2["candidates"]
    assert len(roots) == 1                                  # the exact path sees the same real hits
    assert cands["1"] <= cands["5"] * 1.5                   # the f32 quadratic form and the square tube keep about as much
    if "3" in cands:
        assert cands["1"] <= cands["2"] * 1.2 and cands["2"] <= cands["3"] * 1.01   # looser KU keeps more
        assert cands["4"] <= cands["3"] * 1.05              # same KU, same slack: about as selective
        assert cands["5"] <= cands["3"] * 1.25              # a square tube instead of a slack-inflated cylinder
    if not _ffi.has_crosscheck_modes():                     # the product build refuses the modes it does not carry
        os.environ["RTIOW_SCAN_MODE"] = "3"
        try:
            with pytest.raises(rt.RtiowHipError, match="scan modes 1 and 5"):
                rt.Renderer(0)
        finally:
            os.environ.pop("RTIOW_SCAN_MODE")


def test_tenk_scene_full_size_properties(renderer):
    """BASELINE.json configs[3] (10 001 spheres, 1920x1080) at reduced spp: sharding + pass
    invariance with several bitmap segments per ray in play."""
    flat = rt.random_scene(1, grid=(-50, 49)).flatten()
    renderer.upload_scene(flat)
    w, h, spp = 1920, 1080, 2
    cam = rt.book1_camera(w, h)
    _, whole, st = renderer.render(cam, rt.make_params(w, h, spp))
    acc = np.zeros_like(whole)
    for k in range(4):
        p = rt.make_params(w, h, spp, tile_rows=16, shard_index=k, shard_count=4)
        _, part, _ = renderer.render(cam, p)
        acc[shard_row_map(h, 16, k, 4)] = part
    assert np.array_equal(acc, whole)
    _, p0, _ = renderer.render(cam, rt.make_params(w, h, 1))
    _, p1, _ = renderer.render(cam, rt.make_params(w, h, 1, sample_begin=1))
    assert np.array_equal(p0 + p1, whole)
    assert st["samples"] == w * h * spp
