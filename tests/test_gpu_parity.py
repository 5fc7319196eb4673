"""GPU parity tests: librtiow_hip.so (through its C ABI) against the CPU oracle.

Bar: BIT-EXACT.  The kernel computes every hit decision and every shading value
in the reference's f64 arithmetic (f32 only filters the sphere scan,
conservatively), and pixel sums are exact integers, so the u64 sums must equal
Oracle B's on the same inputs, and the RGBA8 bytes must equal Oracle A's.
The north_star's stated tolerance (per-pixel RMSE < 1e-4 vs CPU) is asserted
against the literal Oracle A as well.
"""
import os
import numpy as np
import pytest

import rtiow_amd as rt

pytestmark = pytest.mark.gpu
RMSE_GATE = 1e-4          # BASELINE.json north_star tolerance


def both(renderer, oracle_mod, flat, w, h, spp, **kw):
    renderer.upload_scene(flat)
    cam = rt.book1_camera(w, h)
    seed = kw.get("seed", 1)
    max_depth = kw.get("max_depth", 50)
    t_min = kw.get("t_min", 1e-4)
    p = rt.make_params(w, h, spp, seed=seed, max_depth=max_depth, t_min=t_min, flags=kw.get("flags", 0))
    sm, fix, st = renderer.render(cam, p)
    op = oracle_mod.make_params(w, h, spp, seed=seed, max_depth=max_depth, t_min=t_min)
    ocam = oracle_mod.camera_from_host(cam)
    fb, sb, stb = oracle_mod.render_b(ocam, flat, op)
    return (sm, fix, st), (fb, sb, stb), (ocam, op)


def test_golden_fixture_bit_exact(renderer, book1_flat, golden_small):
    renderer.upload_scene(book1_flat)
    sm, fix, st = renderer.render(rt.book1_camera(32, 18), rt.make_params(32, 18, 4, seed=1))
    assert np.array_equal(fix, golden_small["fix"])
    assert np.array_equal(sm, golden_small["sum_f32"])
    assert st["rays_traced"] == int(golden_small["rays_b"]) and st["samples"] == 32 * 18 * 4
    assert np.array_equal(renderer.resolve_rgba8(fix, 4, flip=True), golden_small["rgba"])


@pytest.mark.parametrize("w,h,spp,seed", [(64, 36, 4, 1), (400, 225, 10, 1), (200, 133, 7, 0xDEADBEEFCAFE)])
def test_book1_bit_exact_vs_oracle_b(renderer, oracle_mod, book1_flat, w, h, spp, seed):
    """(400,225,10) is BASELINE.json configs[0]; (200,133) is the reference's own 3:2 const size."""
    (sm, fix, st), (fb, sb, stb), _ = both(renderer, oracle_mod, book1_flat, w, h, spp, seed=seed)
    assert np.array_equal(fix, fb)
    assert np.array_equal(sm, sb)
    assert st["rays_traced"] == stb["rays_traced"]
    assert st["samples"] == w * h * spp
    assert st["sphere_tests"] == st["rays_traced"] * len(book1_flat)
    assert np.array_equal(renderer.resolve_rgba8(fix, spp), oracle_mod.resolve_b(fb, spp))


def test_rmse_vs_literal_oracle_a_and_identical_bytes(renderer, oracle_mod, book1_flat):
    w, h, spp = 240, 135, 16
    (sm, fix, st), _, (ocam, op) = both(renderer, oracle_mod, book1_flat, w, h, spp)
    sa, sta = oracle_mod.render_a(ocam, book1_flat, op)
    gpu_mean = fix.astype(np.float64) / 2.0 ** 32 / spp
    rmse = float(np.sqrt(np.mean((gpu_mean - sa / spp) ** 2)))
    assert rmse < 1e-9 < RMSE_GATE
    assert st["rays_traced"] == sta["rays_traced"]
    assert np.array_equal(renderer.resolve_rgba8(fix, spp), oracle_mod.resolve_a(sa, spp))


def test_filter_never_changes_a_result(renderer, oracle_mod, book1_flat):
    """RT_FLAG_NO_FILTER sends every sphere through the exact test: same bits, far more work."""
    (sm, fix, st), (fb, _, _), _ = both(renderer, oracle_mod, book1_flat, 160, 90, 6, flags=rt.RT_FLAG_DIAG_STATS)
    (_, fix2, st2), _, _ = both(renderer, oracle_mod, book1_flat, 160, 90, 6, flags=rt.RT_FLAG_NO_FILTER | rt.RT_FLAG_DIAG_STATS)
    (_, fix3, st3), _, _ = both(renderer, oracle_mod, book1_flat, 160, 90, 6)
    assert np.array_equal(fix3, fix) and st3["candidates"] == 0 and st3["rays_traced"] == st["rays_traced"]
    assert np.array_equal(fix, fix2) and np.array_equal(fix, fb)
    assert st2["candidates"] == st2["sphere_tests"] and st["candidates"] < st["sphere_tests"] // 50
    assert st["exact_roots"] == st2["exact_roots"]


def test_tenk_scene_bit_exact(renderer, oracle_mod):
    """BASELINE.json configs[3]'s scene (10 001 spheres) at a size the oracle finishes in seconds."""
    flat = rt.random_scene(1, grid=(-50, 49)).flatten()
    (sm, fix, st), (fb, sb, stb), _ = both(renderer, oracle_mod, flat, 64, 36, 2)
    assert np.array_equal(fix, fb) and st["rays_traced"] == stb["rays_traced"]
    assert st["n_spheres"] == len(flat) == 10001


@pytest.mark.parametrize("max_depth", [0, 1, 2, 5])
def test_depth_limit(renderer, oracle_mod, book1_flat, max_depth):        # main.rs:40-42
    (sm, fix, st), (fb, _, stb), _ = both(renderer, oracle_mod, book1_flat, 48, 27, 4, max_depth=max_depth)
    assert np.array_equal(fix, fb) and st["rays_traced"] == stb["rays_traced"]
    if max_depth == 0:
        assert not fix.any() and st["rays_traced"] == 0


def test_t_min_is_honoured(renderer, oracle_mod, book1_flat):             # main.rs:44
    (sm, fix, st), (fb, _, _), _ = both(renderer, oracle_mod, book1_flat, 48, 27, 4, t_min=1e-2)
    assert np.array_equal(fix, fb)


def hand_scene(spheres):
    w = rt.HittableList()
    for s in spheres:
        w.push(s)
    return w.flatten()


def test_empty_scene_is_pure_sky(renderer, oracle_mod):
    flat = hand_scene([])
    (sm, fix, st), (fb, _, _), _ = both(renderer, oracle_mod, flat, 40, 30, 3, flags=rt.RT_FLAG_DIAG_STATS)
    assert np.array_equal(fix, fb) and st["rays_traced"] == 40 * 30 * 3 and st["candidates"] == 0


def test_tie_rule_and_material_kinds_on_a_hand_scene(renderer, oracle_mod):
    """Coincident spheres (later wins, mod.rs:61-67), a hollow glass shell (negative radius flips
    the normal through 1/radius, sphere.rs:37), fuzz 0 and 1 metals, a huge ground sphere."""
    flat = hand_scene([
        rt.Sphere(rt.Point3(0, -1000, 0), 1000, rt.Lambertian(rt.Color(0.5, 0.5, 0.5))),
        rt.Sphere(rt.Point3(0, 1, 0), 1.0, rt.Lambertian(rt.Color(0.9, 0.1, 0.1))),
        rt.Sphere(rt.Point3(0, 1, 0), 1.0, rt.Metal(rt.Color(0.8, 0.8, 0.8), 1.0)),      # same sphere, later
        rt.Sphere(rt.Point3(4, 1, 0), 1.0, rt.Dialectric(1.5)),
        rt.Sphere(rt.Point3(4, 1, 0), -0.9, rt.Dialectric(1.5)),
        rt.Sphere(rt.Point3(-4, 1, 0), 1.0, rt.Metal(rt.Color(0.7, 0.6, 0.5), 0.0)),
    ])
    (sm, fix, st), (fb, _, stb), _ = both(renderer, oracle_mod, flat, 120, 68, 8)
    assert np.array_equal(fix, fb) and st["rays_traced"] == stb["rays_traced"]


def test_zero_spp_and_tiny_images(renderer, oracle_mod, book1_flat):
    renderer.upload_scene(book1_flat)
    sm, fix, st = renderer.render(rt.book1_camera(16, 9), rt.make_params(16, 9, 0))
    assert not fix.any() and st["samples"] == 0
    (sm, fix, st), (fb, _, _), _ = both(renderer, oracle_mod, book1_flat, 2, 2, 5)
    assert np.array_equal(fix, fb)


def test_resolve_flip_and_noflip(renderer, oracle_mod, book1_flat):       # main.rs:141-145
    (sm, fix, st), (fb, _, _), _ = both(renderer, oracle_mod, book1_flat, 33, 21, 2)
    a = renderer.resolve_rgba8(fix, 2, flip=False)
    b = renderer.resolve_rgba8(fix, 2, flip=True)
    assert np.array_equal(a[::-1], b) and (a[..., 3] == 255).all()
    assert np.array_equal(a, oracle_mod.resolve_b(fb, 2, flip=False))


def test_f64_divide_and_sqrt_are_correctly_rounded_on_device(renderer):
    rng = np.random.default_rng(5)
    a = np.abs(rng.standard_normal(1 << 18)) * 10.0 ** rng.integers(-30, 30, 1 << 18)
    b = rng.standard_normal(1 << 18) * 10.0 ** rng.integers(-30, 30, 1 << 18)
    a[:4] = [0.0, 1.0, 2.0, np.nextafter(1.0, 2.0)]

    # argument (positive normal >= 2^-767) and the full expansion otherwise: both paths, and the edges between them
    thr = 2.0 ** -767
    special = [np.inf, 5e-324, 2.2250738585072014e-308, np.nextafter(2.2250738585072014e-308, 0.0), thr, np.nextafter(thr, 0.0),
               np.nextafter(thr, 1.0), 1e-300, 1e300, 1.7976931348623157e308, -0.0]
    a[4:4 + len(special)] = special
    a[6400] = thr; a[12800] = np.nextafter(thr, 1.0)
    a[19200] = np.nextafter(thr, 0.0); a[25600] = 0.0; a[32000] = np.inf
    with np.errstate(all="ignore"):
        q, s = renderer.f64_div_sqrt(a, b)
        assert np.array_equal(q, a / b) and np.array_equal(s, np.sqrt(a))
        # negative and NaN arguments: NaN out
        neg = a.copy(); neg[100::977] *= -1.0; neg[50000] = np.nan
        _, s2 = renderer.f64_div_sqrt(neg, b)
        want = np.sqrt(neg)
        assert np.array_equal(np.isnan(s2), np.isnan(want)) and np.array_equal(s2[~np.isnan(want)], want[~np.isnan(want)])


def test_quantize_on_device_is_the_oracles(renderer, oracle_mod):
    """Contract C5 on the device (rt_quantize_device runs the kernel's own quantize()): floor(min(x, 2^16) 2^32) for
    x >= 0, 0 for negatives and NaN -- on the edge values and on 2^18 random radiances of every magnitude."""
    lib = oracle_mod.load()
    rng = np.random.default_rng(9)
    x = np.abs(rng.standard_normal(1 << 18)) * 10.0 ** rng.integers(-40, 12, 1 << 18)
    x[1::7] *= -1.0
    edge = [0.0, -0.0, 1.0, 0.5, 2.0 ** -33, 2.0 ** -32, np.nextafter(2.0 ** -32, 0.0), 1.0 - 2.0 ** -53, 1e30, 2.0 ** 30, 2.0 ** 16,
            np.nextafter(2.0 ** 16, 0.0), np.nextafter(2.0 ** 16, np.inf), np.inf, -np.inf, np.nan, -1.0, 5e-324, 4294967295.75,
            123456.789, 2.0 ** 29 + 2.0 ** -23]
    x[:len(edge)] = edge
    q = renderer.quantize(x)
    want = np.array([lib.oracle_b_quantize(float(v)) for v in x[:4096]], dtype=np.uint64)
    assert np.array_equal(q[:4096], want)
    # the rest against the definition in numpy (exact: scaling by 2^32, truncation)
    xs = np.where(np.isnan(x) | (x < 0), 0.0, np.minimum(x, 2.0 ** 16))
    ref = np.floor(xs).astype(np.uint64) * np.uint64(1 << 32) + np.floor((xs - np.floor(xs)) * 2.0 ** 32).astype(np.uint64)
    assert np.array_equal(q, ref)


def test_errors(renderer, book1_flat):
    fresh = rt.Renderer(0)
    try:
        with pytest.raises(rt.RtiowHipError, match="rt_upload_scene has not been called"):
            fresh.render(rt.book1_camera(8, 8), rt.make_params(8, 8, 1))
        bad = book1_flat.copy()
        bad["kind"][3] = 7
        with pytest.raises(rt.RtiowHipError, match="unknown material kind"):
            fresh.upload_scene(bad)
        # more than RT_MAX_SPHERES (2^24): refused before the list is read (the count alone is enough)
        import ctypes as C
        from rtiow_amd import _ffi
        few = np.ascontiguousarray(book1_flat[:4])
        rc = fresh._lib.rt_upload_scene(fresh._h, few.ctypes.data_as(C.POINTER(_ffi.rt_sphere)), (1 << 24) + 1)
        assert rc != 0 and b"RT_MAX_SPHERES" in fresh._lib.rt_last_error()
        nan = book1_flat.copy()
        nan["center"][2, 0] = np.inf
        with pytest.raises(rt.RtiowHipError, match="finite"):
            fresh.upload_scene(nan)
        fresh.upload_scene(book1_flat)
        with pytest.raises(rt.RtiowHipError, match="65535"):
            fresh.render(rt.book1_camera(70000, 2), rt.make_params(70000, 2, 1))
        with pytest.raises(rt.RtiowHipError, match="t_min"):
            fresh.render(rt.book1_camera(8, 8), rt.make_params(8, 8, 1, t_min=0.0))
    finally:
        fresh.close()
    with pytest.raises(rt.RtiowHipError, match="device_id"):
        rt.Renderer(99)


def _random_world(rng, n, spread, rmin, rmax, ground=True):
    w = rt.HittableList()
    if ground:
        w.push(rt.Sphere(rt.Point3(0, -1000, 0), 1000, rt.Lambertian(rt.Color(0.5, 0.5, 0.5))))
    for _ in range(n):
        c = rng.uniform(-spread, spread, 3)
        c[1] = abs(c[1]) * 0.3
        r = float(rng.uniform(rmin, rmax))
        k = rng.integers(0, 3)
        if k == 0:
            m = rt.Lambertian(rng.uniform(0.05, 0.95, 3))
        elif k == 1:
            m = rt.Metal(rng.uniform(0.5, 1.0, 3), float(rng.uniform(0.0, 0.5)))
        else:
            m = rt.Dialectric(float(rng.uniform(1.2, 2.4)))
        w.push(rt.Sphere(c, r, m))
    return w.flatten()


@pytest.mark.parametrize("case", range(8))
def test_random_scenes_and_cameras_bit_exact(renderer, oracle_mod, case):
    """Scenes the filter constants were NOT tuned on: overlapping spheres of mixed sizes, far-away
    clusters (large |c|), cameras inside the cloud, wide and narrow fields of view, odd image sizes
    (ragged row tiles), several bitmap words per ray."""
    rng = np.random.default_rng(1000 + case)
    n = int(rng.integers(3, 700))
    spread = float(10.0 ** rng.uniform(0.3, 2.5))
    flat = _random_world(rng, n, spread, 0.05 * spread / 10, 0.6 * spread / 10, ground=bool(case % 2))
    w, h, spp = int(rng.integers(17, 90)), int(rng.integers(9, 60)), int(rng.integers(1, 6))
    look_from = rng.uniform(-spread, spread, 3); look_from[1] = abs(look_from[1]) * 0.3 + 0.5
    look_at = rng.uniform(-spread, spread, 3) * 0.3
    cam = rt.Camera(look_from, look_at, rt.Vec3(0, 1, 0), float(rng.uniform(10, 100)), w / h,
                    float(rng.uniform(0.0, 0.3)), float(np.linalg.norm(look_from - look_at)))
    seed = int(rng.integers(1, 2 ** 62))
    renderer.upload_scene(flat)
    sm, fix, st = renderer.render(cam, rt.make_params(w, h, spp, seed=seed, tile_rows=int(rng.integers(1, 9))))
    fb, sb, stb = oracle_mod.render_b(oracle_mod.camera_from_host(cam), flat, oracle_mod.make_params(w, h, spp, seed=seed))
    assert np.array_equal(fix, fb), case
    assert st["rays_traced"] == stb["rays_traced"]


@pytest.mark.parametrize("w,h,spp", [(64, 36, 37), (48, 27, 100), (33, 19, 255), (31, 17, 256), (29, 16, 257), (23, 13, 600), (75, 41, 17), (64, 36, 18),
                                     (51, 29, 31), (97, 55, 5), (90, 51, 8), (83, 47, 9), (80, 45, 10), (71, 40, 13), (64, 36, 16), (57, 33, 36)])
def test_block_sums_in_lds_bit_exact(renderer, oracle_mod, book1_flat, w, h, spp):
    """A wave keeps the sums of its work blocks (256 consecutive pixel-samples, pixel-major) in LDS and writes a block to the frame
    buffer once.  A block's pixels must fit the ring's pixel slots -- 16 on the shipped scan mode's kernels, 8 on the others --:
    blocks of 256 from 37 / 17 samples per pixel on, of 192, 128 or 64 below that, down to 9 / 5 samples per pixel.  Sizes chosen so that
    blocks start and end in the middle of pixels and rows (spp not a divisor of 256, odd widths, a last
    block of fewer than 256 items)."""
    (sm, fix, st), (fb, sb, stb), _ = both(renderer, oracle_mod, book1_flat, w, h, spp)
    assert np.array_equal(fix, fb) and np.array_equal(sm, sb)
    assert st["rays_traced"] == stb["rays_traced"] and st["samples"] == w * h * spp
    if spp >= (17 if st["scan_mode"] == 5 else 37):
        assert st["direct_samples"] < st["samples"] // 100       # (orphans of long paths only)
    elif spp >= (5 if st["scan_mode"] == 5 else 9):
        assert st["direct_samples"] < st["samples"] // 3         # (small blocks leave the ring sooner: paths of more than a few bounces)
    else:
        assert st["direct_samples"] == st["samples"]             # (the cross-check scan modes run on the general kernel: 8 pixel slots)


def test_small_spp_goes_to_the_frame_buffer_directly(renderer, oracle_mod, book1_flat):
    (sm, fix, st), (fb, _, _), _ = both(renderer, oracle_mod, book1_flat, 80, 45, 4)
    assert np.array_equal(fix, fb) and st["direct_samples"] == st["samples"] == 80 * 45 * 4


def test_ring_minimum_follows_the_kernel(oracle_mod, book1_flat):
    """6 samples per pixel: block sums in LDS on the shipped scan mode's kernels (16 pixel slots: 64 consecutive samples touch 12 pixels) --
    small grid, and the general kernel: the same scene without its grid (RTIOW_NO_GRID=1) --, straight to the frame buffer on the
    unfiltered scan's kernel (RT_FLAG_NO_FILTER; 8 slots: from 9 on); the same frame."""
    w, h, spp = 80, 45, 6
    cam = rt.book1_camera(w, h)
    fb, _, _ = oracle_mod.render_b(oracle_mod.camera_from_host(cam), book1_flat, oracle_mod.make_params(w, h, spp))
    with rt.Renderer(0) as r:
        r.upload_scene(book1_flat)
        _, fix, st = r.render(cam, rt.make_params(w, h, spp))
    if st["scan_mode"] == 5:
        assert st["kernel_variant"] & 1 and st["direct_samples"] < st["samples"] // 3
    assert np.array_equal(fix, fb)
    os.environ["RTIOW_NO_GRID"] = "1"
    try:
        with rt.Renderer(0) as r:
            r.upload_scene(book1_flat)
            _, fix2, st2 = r.render(cam, rt.make_params(w, h, spp))
    finally:
        os.environ.pop("RTIOW_NO_GRID")
    assert not (st2["kernel_variant"] & 1) and np.array_equal(fix2, fb)
    if st2["scan_mode"] == 5:
        assert st2["direct_samples"] < st2["samples"] // 3
    with rt.Renderer(0) as r:
        r.upload_scene(book1_flat)
        _, fix3, st3 = r.render(cam, rt.make_params(w, h, spp, flags=rt.RT_FLAG_NO_FILTER))
    assert st3["scan_mode"] == 0 and st3["direct_samples"] == st3["samples"] and np.array_equal(fix3, fb)


def test_hall_of_mirrors_runs_paths_to_the_depth_limit(renderer, oracle_mod):
    """The camera between two large fuzz-0 metal spheres, a glass and a diffuse sphere around: 6.4 rays per
    sample on average and paths that run out of depth (main.rs:40-42) -- blocks of work are held open by
    single long paths for many passes.  (Blocks held open long enough to lose their LDS entry -- "orphans" --
    need a short AVERAGE path beside the long ones: test_gpu_properties.py checks them on configs[1].)"""
    flat = hand_scene([
        rt.Sphere(rt.Point3(0, -1000, 0), 1000, rt.Metal(rt.Color(0.9, 0.9, 0.9), 0.0)),
        rt.Sphere(rt.Point3(0, 1002.5, 0), 1000, rt.Metal(rt.Color(0.95, 0.95, 0.95), 0.0)),
        rt.Sphere(rt.Point3(0, 1, 0), 1.0, rt.Dialectric(1.5)),
        rt.Sphere(rt.Point3(-4, 1, 0), 1.0, rt.Lambertian(rt.Color(0.4, 0.2, 0.1))),
        rt.Sphere(rt.Point3(4, 1, 0), 1.0, rt.Metal(rt.Color(0.7, 0.6, 0.5), 0.0)),
    ])
    (sm, fix, st), (fb, _, stb), _ = both(renderer, oracle_mod, flat, 96, 54, 64)
    assert np.array_equal(fix, fb) and st["rays_traced"] == stb["rays_traced"]
    assert stb["end_depth"] > 0 and st["rays_traced"] > 6 * st["samples"]


def test_tie_between_spheres_of_different_tiles(renderer, oracle_mod):
    """Coincident spheres whose list positions are more than one 32-sphere filter tile apart (indices 3 and
    70, and a third copy at 140): the later one must win the tie (mod.rs:61-67) although its candidates reach
    the pooled exact tests in different rounds."""
    spheres = [rt.Sphere(rt.Point3(0, -1000, 0), 1000, rt.Lambertian(rt.Color(0.5, 0.5, 0.5)))]
    rng = np.random.default_rng(7)
    for k in range(1, 160):
        if k in (3, 70, 140):
            m = [rt.Lambertian(rt.Color(0.9, 0.1, 0.1)), rt.Metal(rt.Color(0.1, 0.9, 0.1), 0.3), rt.Lambertian(rt.Color(0.1, 0.1, 0.9))][(3, 70, 140).index(k)]
            spheres.append(rt.Sphere(rt.Point3(0, 1, 0), 1.0, m))
        else:
            c = rng.uniform(-9, 9, 3); c[1] = 0.2
            spheres.append(rt.Sphere(c, 0.2, rt.Lambertian(rng.uniform(0.1, 0.9, 3))))
    flat = hand_scene(spheres)
    (sm, fix, st), (fb, _, stb), _ = both(renderer, oracle_mod, flat, 120, 68, 40)
    assert np.array_equal(fix, fb) and st["rays_traced"] == stb["rays_traced"]
    # the blue copy (index 140) is what the camera sees on the big sphere: its centre pixel is blue-ish
    mean = fix.astype(np.float64) / 2.0 ** 32 / 40
    cy, cx = 43, 60                                             # (rows count from the bottom: j = 43 is the middle of the big sphere)
    assert mean[cy, cx, 2] > mean[cy, cx, 0]


def test_zero_radius_is_rejected(book1_flat):
    fresh = rt.Renderer(0)
    try:
        bad = book1_flat.copy()
        bad["radius"][5] = 0.0
        with pytest.raises(rt.RtiowHipError, match="radius must not be zero"):
            fresh.upload_scene(bad)
        with pytest.raises(rt.RtiowHipError, match="rt_upload_scene has not been called"):     # nothing half-uploaded
            fresh.render(rt.book1_camera(8, 8), rt.make_params(8, 8, 1))
    finally:
        fresh.close()


@pytest.mark.parametrize("w,h,spp,tile", [(40000, 2, 3, 1), (2, 40000, 3, 33000), (35000, 3, 40, 2), (3, 1000, 300, 7)])
def test_extreme_aspect_ratios_and_tile_sizes(renderer, oracle_mod, book1_flat, w, h, spp, tile):
    """Item -> (pixel, sample, row, column) runs on multiply-high with host-made reciprocals (udiv_small): widths and
    tile heights beyond 2^15 take the 'quotient is 0 or 1' branch, 3-pixel rows carry a block over many rows."""
    renderer.upload_scene(book1_flat)
    cam = rt.book1_camera(w, h)
    sm, fix, st = renderer.render(cam, rt.make_params(w, h, spp, seed=3, tile_rows=tile))
    fb, sb, stb = oracle_mod.render_b(oracle_mod.camera_from_host(cam), book1_flat, oracle_mod.make_params(w, h, spp, seed=3))
    assert np.array_equal(fix, fb) and st["rays_traced"] == stb["rays_traced"] and st["samples"] == w * h * spp
    if tile < h:                                            # and as two shards
        full = np.zeros_like(fix)
        for k in range(2):
            p = rt.make_params(w, h, spp, seed=3, tile_rows=tile, shard_index=k, shard_count=2)
            _, part, _ = renderer.render(cam, p)
            full[rt.shard_row_indices(p)] = part
        assert np.array_equal(full, fb)


def _grid_scene(name):
    """Scenes that put the position-tiled scan in its corners: a slab as thick as the scene, clumps whose
    cells overflow into the global tiles, a line of spheres (no extent in z), a hundred spheres on one point, and a
    scene of 3 000 spheres (a grid of more than 64 cells: the LDS form of the tile list)."""
    rng = np.random.default_rng(23)
    def sph(c, r):
        flat = np.zeros(len(r), dtype=rt.SPHERE_DTYPE)
        flat["center"], flat["radius"] = c, r
        flat["kind"] = rng.integers(0, 3, len(r))
        flat["albedo"] = rng.uniform(0.2, 0.9, (len(r), 3))
        flat["param"] = np.where(flat["kind"] == 2, 1.5, rng.uniform(0.0, 0.4, len(r)))
        return flat
    if name == "cloud":
        return sph(rng.uniform(-6, 6, (900, 3)), rng.uniform(0.1, 0.4, 900))
    if name == "clusters":
        c = np.concatenate([rng.normal((-3, 0.3, -2), 0.5, (300, 3)), rng.normal((3, 0.3, 1), 0.3, (250, 3)),
                            rng.uniform(-8, 8, (150, 3)) * (1, 0.02, 1)])
        return sph(c, rng.uniform(0.05, 0.2, len(c)))
    if name == "line":
        return sph(np.stack([np.linspace(-10, 10, 300), np.full(300, 0.2), np.zeros(300)], 1), np.where(np.arange(300) % 50 == 0, 1.5, 0.2))
    if name == "same":
        return sph(np.tile([[0.0, 0.5, 0.0]], (100, 1)), np.linspace(0.1, 0.6, 100))
    if name == "many":
        c = rng.uniform(-30, 30, (3000, 3)) * (1, 0.01, 1) + (0, 0.3, 0)
        return sph(c, rng.uniform(0.1, 0.3, 3000))
    raise KeyError(name)


@pytest.mark.parametrize("name", ["cloud", "clusters", "line", "same", "many"])
def test_position_tiled_scan_bit_exact_on_awkward_layouts(renderer, oracle_mod, name):
    flat = _grid_scene(name)
    for w, h, spp in [(48, 27, 3), (24, 13, 40)]:
        (sm, fix, st), (fb, sb, stb), _ = both(renderer, oracle_mod, flat, w, h, spp)
        assert np.array_equal(fix, fb), name
        assert st["rays_traced"] == stb["rays_traced"]
    # and the filter decides nothing here either
    (sm2, fix2, st2), _, _ = both(renderer, oracle_mod, flat, 48, 27, 3, flags=rt.RT_FLAG_NO_FILTER)
    (sm, fix, st), _, _ = both(renderer, oracle_mod, flat, 48, 27, 3)
    assert np.array_equal(fix, fix2)


@pytest.mark.parametrize("scale,cam_far,t_min,n", [(1e11, 2e4, 1e-4, 150), (1e11, 2e4, 1e-4, 3000), (1e-13, 1.0, 1e-17, 150)])
def test_rays_and_scenes_outside_the_filters_analysed_range(renderer, oracle_mod, scale, cam_far, t_min, n):
    """A camera 2.7e16 away from a scene of size 1e12 (|o| >= 1e15: the f32 filter and the grid footprint both answer
    "cannot tell" for the camera rays, which then test every sphere exactly while their wave scans every tile; the
    bounce rays are back inside the range), and a scene of size 1e-12 (radii squared below 1e-30, directions squared
    below 1e-20): the same bits as the oracle."""
    rng = np.random.default_rng(5)
    half = 8.0 if n < 1000 else 40.0                    # (3000 spheres: a grid of more than 64 cells)
    flat = np.zeros(n + 1, dtype=rt.SPHERE_DTYPE)
    flat["center"][0], flat["radius"][0] = (0.0, -1000.0 * scale, 0.0), 1000.0 * scale
    flat["center"][1:] = rng.uniform(-half, half, (n, 3)) * (1, 0.0, 1) * scale + (0, 0.3 * scale, 0)
    flat["radius"][1:] = rng.uniform(0.1, 0.3, n) * scale
    flat["kind"] = rng.integers(0, 3, n + 1)
    flat["kind"][0] = 0
    flat["albedo"] = rng.uniform(0.3, 0.9, (n + 1, 3))
    flat["param"] = np.where(flat["kind"] == 2, 1.5, 0.2)
    renderer.upload_scene(flat)
    w, h, spp = 40, 24, 3
    far = scale * cam_far
    cam = rt.Camera(rt.Point3(13 * far, 2 * far, 3 * far), rt.Point3(0, 0, 0), rt.Vec3(0, 1, 0), 25.0 / cam_far, w / h, 0.1 * scale, float(np.sqrt(182.0)) * far)
    # (t_min is a ray PARAMETER: camera rays have |d| of the camera's distance, scattered rays |d| of order 1)
    sm, fix, st = renderer.render(cam, rt.make_params(w, h, spp, t_min=t_min))
    fb, sb, stb = oracle_mod.render_b(oracle_mod.camera_from_host(cam), flat, oracle_mod.make_params(w, h, spp, t_min=t_min))
    assert np.array_equal(fix, fb)
    assert st["rays_traced"] == stb["rays_traced"] and st["rays_traced"] > 1.5 * w * h * spp     # (the scene is hit)


@pytest.mark.gpu
@pytest.mark.parametrize("scene,w,h,spp,flags", [("book", 160, 90, 150, 0), ("book", 96, 54, 300, rt.RT_FLAG_UNIFORM53),
                                                  ("3000", 120, 68, 147, 0), ("book", 131, 57, 211, 0), ("book", 150, 80, 69, 0),
                                                  ("3000", 97, 61, 100, 0)])
def test_large_work_blocks_bit_exact(oracle_mod, book1_flat, scene, w, h, spp, flags):
    """Launches of >= 2 x 10^8 pixel-samples at >= 69 samples per pixel hand out work in blocks of 1 024 pixel-samples instead of
    256 (another instantiation of the kernel, whose ring of block sums is 2 blocks x 16 pixels instead of 4 x 8: rt_stats.kernel_variant bit 2).  RTIOW_LARGE_BLOCK_MIN_ITEMS=0 selects it for a
    launch small enough for the oracle: both grid variants, the 53-bit stream, ragged sizes (a last block that is not full,
    blocks that straddle rows), and a second pass with sample_begin -- against Oracle B, bit for bit."""
    if flags & rt.RT_FLAG_UNIFORM53 and os.environ.get("RTIOW_SCAN_MODE", "5") != "5":
        pytest.skip("RT_FLAG_UNIFORM53 runs with the shipped scan mode only")
    flat = book1_flat if scene == "book" else rt.random_scene(3, grid=(-27, 27)).flatten()
    os.environ["RTIOW_LARGE_BLOCK_MIN_ITEMS"] = "0"
    try:
        with rt.Renderer(0) as r:
            r.upload_scene(flat)
            cam = rt.book1_camera(w, h)
            _, fix, st = r.render(cam, rt.make_params(w, h, spp, flags=flags))
            # (only the shipped scan mode has the large-block instantiations: the cross-check modes run this test on blocks of 256)
            # ... from 147 samples per pixel on; the small-grid kernel (kernel_variant bit 0), whose large-block ring has 16 pixel slots, from 69 on
            large150 = 4 if st["scan_mode"] == 5 else 0
            large = large150 if (spp >= 147 or (st["kernel_variant"] & 1 and spp >= 69)) else 0
            assert (scene == "book") == bool(st["kernel_variant"] & 1) or st["scan_mode"] != 5
            assert (st["kernel_variant"] & 4) == large and st["direct_samples"] < st["samples"] // 100
            ocam = oracle_mod.camera_from_host(cam)
            fb, _, stb = oracle_mod.render_b(ocam, flat, oracle_mod.make_params(w, h, spp, uniform53=bool(flags & rt.RT_FLAG_UNIFORM53), nthreads=8))
            assert np.array_equal(fix, fb) and st["rays_traced"] == stb["rays_traced"]
            _, fix2, st2 = r.render(cam, rt.make_params(w, h, 150, sample_begin=1000, flags=flags))
            fb2, _, _ = oracle_mod.render_b(ocam, flat, oracle_mod.make_params(w, h, 150, sample_begin=1000, uniform53=bool(flags & rt.RT_FLAG_UNIFORM53), nthreads=8))
            assert (st2["kernel_variant"] & 4) == large150 and np.array_equal(fix2, fb2)
        os.environ["RTIOW_LARGE_BLOCK_MIN_ITEMS"] = str(1 << 62)          # never: the same launch on blocks of 256 gives the same frame
        with rt.Renderer(0) as r:
            r.upload_scene(flat)
            _, fix3, st3 = r.render(rt.book1_camera(w, h), rt.make_params(w, h, spp, flags=flags))
            assert not (st3["kernel_variant"] & 4) and np.array_equal(fix3, fb)
    finally:
        os.environ.pop("RTIOW_LARGE_BLOCK_MIN_ITEMS")
