"""GPU vs Oracle B (bit-exact) on what round 2 only ever compared GPU-with-GPU or not at all:

  * additive passes: sample_begin != 0 and RT_FLAG_ACCUMULATE against ONE oracle render (main.rs:130-137 is the loop
    the passes split), with and without the per-wave LDS block sums (spp >= 37 / spp < 37);
  * BASELINE configs[1] (1200x675x100) and the bench headline (1200x675x500) at their true size, row by row;
  * the knobs that select other code paths of the shipped kernel (grid resolution, no grid, blocks per CU, block sums off)
    -- each against the oracle, not against the default GPU frame;
  * degenerate rays: a zero-length or underflowing direction makes sphere.rs:28-34 produce NaN / infinite roots, which
    `root < t_min || t_max < root` ACCEPTS (mod.rs:61-67 then takes every later sphere too); the kernel follows the
    reference there (rays outside the filter's analysed range run HittableList::hit as written);
  * bench.py's N > 1 path, rehearsed with two ranks on this one GPU, against the N = 1 frame.
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import rtiow_amd as rt
from rtiow_amd import _ffi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def oracle_frame(oracle_mod, flat, cam, w, h, spp, **kw):
    """Oracle B's exact sums and stats for a host-mirror camera."""
    fb, _, stb = oracle_mod.render_b(oracle_mod.camera_from_host(cam), flat, oracle_mod.make_params(w, h, spp, **kw))
    return fb, stb


# ---- additive passes ---------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("begin,spp", [(37, 63), (5, 20), (9, 12), (3, 7), (11, 4), (1000003, 41)])
def test_sample_begin_vs_oracle(renderer, oracle_mod, book1_flat, begin, spp):
    """One launch that starts at sample `begin`: (37, 63) keeps block sums in LDS on work blocks of 256 pixel-samples, (5, 20) and
    (9, 12) on smaller blocks (192 / 128 / 64: what fits the ring's pixel slots), (3, 7) only on the shipped scan mode's kernels (16 pixel slots:
    from 5 spp on; the others' 8: from 9), (11, 4) sends every sample to the frame buffer directly, the last has a large first index."""
    w, h = 160, 90
    renderer.upload_scene(book1_flat)
    cam = rt.book1_camera(w, h)
    _, fix, st = renderer.render(cam, rt.make_params(w, h, spp, sample_begin=begin))
    fb, stb = oracle_frame(oracle_mod, book1_flat, cam, w, h, spp, sample_begin=begin)
    assert np.array_equal(fix, fb) and st["rays_traced"] == stb["rays_traced"]
    assert (st["direct_samples"] == st["samples"]) == (spp < (5 if st["scan_mode"] == 5 else 9))


@pytest.mark.parametrize("passes", [((0, 60), (60, 40)), ((0, 10), (10, 27), (37, 63)), ((0, 99), (99, 1))])
def test_accumulate_flag_passes_equal_one_oracle_render(renderer, oracle_mod, book1_flat, passes):
    """Two or three launches into ONE device buffer with RT_FLAG_ACCUMULATE == a single Oracle-B render of all 100
    samples (u64 sums and the RGBA8 bytes of the resolve)."""
    torch = pytest.importorskip("torch")
    w, h, total = 200, 112, 100
    renderer.upload_scene(book1_flat)
    cam = rt.book1_camera(w, h)
    d_fix = torch.full((h, w, 3), 12345, dtype=torch.int64, device="cuda:0")      # (the first launch must clear it)
    stream = torch.cuda.current_stream().cuda_stream
    rays = 0
    for k, (begin, n) in enumerate(passes):
        renderer.render_device(cam, rt.make_params(w, h, n, sample_begin=begin, flags=rt.RT_FLAG_ACCUMULATE if k else 0),
                               d_fix.data_ptr(), stream)
        rays += renderer.last_stats()["rays_traced"]
    d_rgba = torch.empty((h, w, 4), dtype=torch.uint8, device="cuda:0")
    renderer.resolve_rgba8_device(d_fix.data_ptr(), w, h, total, 1, d_rgba.data_ptr(), stream)
    torch.cuda.synchronize()
    fb, stb = oracle_frame(oracle_mod, book1_flat, cam, w, h, total)
    assert np.array_equal(d_fix.cpu().numpy().view(np.uint64), fb) and rays == stb["rays_traced"]
    assert np.array_equal(d_rgba.cpu().numpy(), oracle_mod.resolve_b(fb, total, flip=True))


# ---- configs[1] and the bench headline at full size, row by row ------------------------------------------------------------

@pytest.mark.parametrize("spp,rows", [(100, (3, 337)), (500, (120, 674))])
def test_cfg2_and_target_rows_bit_exact_vs_oracle(renderer, oracle_mod, book1_flat, spp, rows):
    """1200x675 at 100 spp (BASELINE configs[1]) and at 500 spp (the configuration bench.py times): image rows rendered
    alone (tile_rows = 1, shard_count = H: same global pixel keys and samples as in the full frame) against Oracle B, and
    the same rows of a full-frame launch."""
    from test_gpu_full_configs import probe_row
    w, h = 1200, 675
    renderer.upload_scene(book1_flat)
    probed = {}
    for j in rows:
        g, o, rg, ro = probe_row(renderer, oracle_mod, book1_flat, w, h, spp, j)
        assert np.array_equal(g, o), f"row {j}"
        assert rg == ro, f"row {j}"
        probed[j] = g
    _, full, st = renderer.render(rt.book1_camera(w, h), rt.make_params(w, h, spp))
    assert st["samples"] == w * h * spp
    for j, g in probed.items():
        assert np.array_equal(full[j], g), f"row {j} of the full frame"


# ---- the knobs that select other code paths --------------------------------------------------------------------------------

KNOBS = [{"RTIOW_BLOCKS_PER_CU": "1"}, {"RTIOW_BLOCKS_PER_CU": "2"}, {"RTIOW_RING_MIN_SPP": "1000"}, {"RTIOW_NO_GRID": "1"},
         {"RTIOW_GRID_DIM": "1"}, {"RTIOW_GRID_DIM": "3"}, {"RTIOW_GRID_DIM": "8"}, {"RTIOW_GRID_DIM": "42"},
         {"RTIOW_GRID_DIM": "9", "RTIOW_BLOCKS_PER_CU": "3", "RTIOW_RING_MIN_SPP": "41"}]


@pytest.fixture(scope="module")
def knob_scenes(oracle_mod, book1_flat):
    """(flat, w, h, spp, oracle sums, oracle rays) for the book scene and a 3 000-sphere scene; 40 spp: block sums in LDS
    by default (>= 37), off under RTIOW_RING_MIN_SPP=1000."""
    out = {}
    mid = rt.random_scene(1, grid=(-27, 27)).flatten()
    assert 2900 < len(mid) < 3100
    for name, flat, w, h, spp in (("book", book1_flat, 240, 135, 40), ("3k", mid, 128, 72, 40)):
        fb, stb = oracle_frame(oracle_mod, flat, rt.book1_camera(w, h), w, h, spp)
        out[name] = (flat, w, h, spp, fb, stb["rays_traced"])
    return out


@pytest.mark.parametrize("scene", ["book", "3k"])
@pytest.mark.parametrize("env", KNOBS, ids=lambda e: ",".join(f"{k[6:]}={v}" for k, v in e.items()))
def test_live_knobs_vs_oracle(knob_scenes, scene, env):
    flat, w, h, spp, fb, rays = knob_scenes[scene]
    os.environ.update(env)
    try:
        r = rt.Renderer(0)                      # RTIOW_BLOCKS_PER_CU / RTIOW_RING_MIN_SPP are read here,
        r.upload_scene(flat)                    # RTIOW_NO_GRID / RTIOW_GRID_DIM here
        _, fix, st = r.render(rt.book1_camera(w, h), rt.make_params(w, h, spp))
        r.close()
    finally:
        for k in env:
            os.environ.pop(k)
    assert np.array_equal(fix, fb), env
    assert st["rays_traced"] == rays
    if "RTIOW_RING_MIN_SPP" in env:
        assert (st["direct_samples"] == st["samples"]) == (int(env["RTIOW_RING_MIN_SPP"]) > spp)
    if env.get("RTIOW_BLOCKS_PER_CU"):
        assert st["grid_blocks"] <= 256 * int(env["RTIOW_BLOCKS_PER_CU"])
    if env.get("RTIOW_GRID_DIM") == "42" or "RTIOW_NO_GRID" in env:
        assert st["kernel_variant"] == 0        # > 64 cells / no grid: the general instantiation
    if scene == "book" and env.get("RTIOW_GRID_DIM") in ("1", "3"):
        assert st["kernel_variant"] == 1        # the small-grid instantiation (the 3 000-sphere scene overflows such grids: no grid)


# ---- degenerate rays -------------------------------------------------------------------------------------------------------

def degenerate_camera(oracle_mod, scale):
    """A camera whose every ray has direction `scale` x (something of order 1): lower_left_corner = origin and a viewport
    of size `scale`.  scale = 0: the direction is the zero vector (a = 0, half_b = 0, disc = 0, root = 0/0 = NaN for EVERY
    sphere); scale = 1e-170: a = |d|^2 underflows to 0 while d does not (roots of -inf, +inf or NaN)."""
    c = _ffi.rt_camera()
    c.origin = (C.c_double * 3)(13.0, 2.0, 3.0)
    c.lower_left_corner = (C.c_double * 3)(13.0, 2.0, 3.0)
    c.horizontal = (C.c_double * 3)(scale * -0.3, 0.0, scale * 1.0)
    c.vertical = (C.c_double * 3)(scale * -0.1, scale * 1.0, scale * -0.05)
    c.u = (C.c_double * 3)(0.0, 0.0, 1.0)
    c.v = (C.c_double * 3)(0.0, 1.0, 0.0)
    c.lens_radius = 0.0
    oc = oracle_mod.camera()
    for name, _ in _ffi.rt_camera._fields_:
        setattr(oc, name, getattr(c, name))
    return c, oc


@pytest.mark.parametrize("scale", [0.0, 1e-170])
@pytest.mark.parametrize("flags", [0, rt.RT_FLAG_NO_FILTER])
def test_degenerate_directions_follow_the_reference(renderer, oracle_mod, scale, flags):
    """sphere.rs:29-33 accepts a NaN root (both comparisons are false), mod.rs:63-64 then carries closest_so_far = NaN and
    every later sphere with a root >= t_min is accepted as well; the path goes on through NaN points until the depth
    limit.  Oracle B restates exactly that; the GPU must give the same sums AND the same ray counts.  (What the image
    shows is black: a NaN radiance quantises to 0, as the reference's `NaN as u8`.)"""
    flat = rt.random_scene(1, grid=(-4, 4)).flatten()              # ~80 small spheres + ground + the three big ones
    w, h, spp = 24, 14, 3
    cam, ocam = degenerate_camera(oracle_mod, scale)
    renderer.upload_scene(flat)
    _, fix, st = renderer.render(cam, rt.make_params(w, h, spp, max_depth=12, flags=flags))
    fb, _, stb = oracle_mod.render_b(ocam, flat, oracle_mod.make_params(w, h, spp, max_depth=12))
    assert np.array_equal(fix, fb)
    assert st["rays_traced"] == stb["rays_traced"]
    if scale == 0.0:                            # every scan ends on a NaN hit: no path leaves before the depth limit
        assert st["rays_traced"] == w * h * spp * 12 and not fix.any()


def test_rays_inside_and_outside_the_filters_range_in_one_wave(renderer, oracle_mod, book1_flat):
    """Every ray points from the camera at the scene's centre, with length 182^0.5 (1e-12 + 1e-10 u): |d|^2 <= 1e-20 --
    outside the filter's analysed range -- for the first few columns only, so the leftmost waves mix rays that run
    HittableList::hit as written with rays that go through the filter."""
    c = _ffi.rt_camera()
    c.origin = (C.c_double * 3)(13.0, 2.0, 3.0)
    c.lower_left_corner = (C.c_double * 3)(13.0 - 13e-12, 2.0 - 2e-12, 3.0 - 3e-12)
    c.horizontal = (C.c_double * 3)(-13e-10, -2e-10, -3e-10)
    c.vertical = (C.c_double * 3)(0.0, 0.0, 0.0)
    c.u = (C.c_double * 3)(0.0, 0.0, 1.0)
    c.v = (C.c_double * 3)(0.0, 1.0, 0.0)
    c.lens_radius = 0.0
    oc = oracle_mod.camera()
    for name, _ in _ffi.rt_camera._fields_:
        setattr(oc, name, getattr(c, name))
    w, h, spp = 64, 40, 4
    renderer.upload_scene(book1_flat)
    _, fix, st = renderer.render(c, rt.make_params(w, h, spp))
    fb, _, stb = oracle_mod.render_b(oc, book1_flat, oracle_mod.make_params(w, h, spp))
    assert np.array_equal(fix, fb) and st["rays_traced"] == stb["rays_traced"]
    assert fix.any()


# ---- bench.py --gpus 2, rehearsed on this one GPU ----------------------------------------------------------------------------

def _bench(*args):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


def test_bench_two_ranks_rehearsed_on_one_gpu_give_the_single_gpu_frame():
    """bench.py's N > 1 path (self-launch of the ranks as a child, interleaved row shards, the gather, reassembly, resolve)
    with both ranks on cuda:0 and the gather over gloo: exit code 0, n_gpus 2, and the CRC of the assembled exact sums equals
    the N = 1 CRC of the same 320x180x80 frame."""
    two = _bench("--gpus", "2", "--rehearse-on-one-gpu", "--width", "320", "--height", "180", "--spp", "40", "--steps", "2",
                 "--warmup", "1", "--no-cpu-baseline")
    one = _bench("--width", "320", "--height", "180", "--spp", "80", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                 "--no-other-configs")
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    assert two["config"]["spp"] == one["config"]["spp"] == 80
    assert two["config"]["frame_crc32"] == one["config"]["frame_crc32"] is not None
    assert len(two["per_rank"]["kernel_ms"]) == 2 and all(v > 0 for v in two["per_rank"]["kernel_ms"])
    assert two["scaling"] == "weak" and two["value"] > 0 and one["rmse_vs_cpu"] is None


# ---- material / geometry corners the reference allows and no other GPU test reaches ------------------------------------------

def _hand(spheres):
    w = rt.HittableList()
    for s in spheres:
        w.push(s)
    return w.flatten()


@pytest.mark.parametrize("case", ["inside_lambertian", "inside_metal_shell", "fuzz_above_one", "ir_below_and_at_one", "nested_glass"])
def test_reference_corners_on_hand_scenes(renderer, oracle_mod, case):
    """* the camera INSIDE a Lambertian / a Metal sphere: every hit is a back-face hit (mod.rs:21-22 flips the normal), the
         scattered ray leaves the surface inwards and meets its own sphere again from inside (the far root, sphere.rs:30);
       * Metal::new does not clamp fuzz (materials.rs:39-46): fuzz 3 throws most reflections below the surface (absorbed,
         materials.rs:57-61);
       * Dialectric with ir 0.5 (total internal reflection from OUTSIDE, refraction_ratio = 2) and ir 1.0 (no bending,
         Schlick r0 = 0);
       * a glass sphere inside a glass sphere inside a glass sphere (front / back faces alternate, long paths)."""
    ground = rt.Sphere(rt.Point3(0, -1000, 0), 1000, rt.Lambertian(rt.Color(0.5, 0.5, 0.5)))
    if case == "inside_lambertian":       # book camera at (13, 2, 3): enclosed by a sphere of radius 6 around it
        flat = _hand([ground, rt.Sphere(rt.Point3(13, 2, 3), 6.0, rt.Lambertian(rt.Color(0.8, 0.7, 0.6))),
                      rt.Sphere(rt.Point3(10, 1, 2), 1.0, rt.Metal(rt.Color(0.9, 0.9, 0.9), 0.1))])
    elif case == "inside_metal_shell":
        flat = _hand([rt.Sphere(rt.Point3(13, 2, 3), 8.0, rt.Metal(rt.Color(0.9, 0.8, 0.7), 0.05)),
                      rt.Sphere(rt.Point3(9, 1, 2), 1.0, rt.Lambertian(rt.Color(0.2, 0.6, 0.9))),
                      rt.Sphere(rt.Point3(11, 3, 5), 0.7, rt.Dialectric(1.5))])
    elif case == "fuzz_above_one":
        flat = _hand([ground] + [rt.Sphere(rt.Point3(2.2 * k, 1, 0), 1.0, rt.Metal(rt.Color(0.8, 0.8, 0.8), f))
                                 for k, f in zip(range(-2, 3), (0.0, 1.0, 3.0, 10.0, 1.5))])
    elif case == "ir_below_and_at_one":
        flat = _hand([ground, rt.Sphere(rt.Point3(-2.5, 1, 0), 1.0, rt.Dialectric(0.5)), rt.Sphere(rt.Point3(0, 1, 0), 1.0, rt.Dialectric(1.0)),
                      rt.Sphere(rt.Point3(2.5, 1, 0), 1.0, rt.Dialectric(1.0 / 1.5)), rt.Sphere(rt.Point3(5, 1, 0), 1.0, rt.Dialectric(4.0))])
    else:
        flat = _hand([ground, rt.Sphere(rt.Point3(0, 1.5, 0), 1.5, rt.Dialectric(1.5)), rt.Sphere(rt.Point3(0, 1.5, 0), 1.0, rt.Dialectric(1.3)),
                      rt.Sphere(rt.Point3(0, 1.5, 0), 0.5, rt.Dialectric(2.0)), rt.Sphere(rt.Point3(0, 1.5, 0), -0.45, rt.Dialectric(2.0))])
    w, h, spp = 96, 54, 40
    cam = rt.book1_camera(w, h)
    renderer.upload_scene(flat)
    _, fix, st = renderer.render(cam, rt.make_params(w, h, spp))
    fb, _, stb = oracle_mod.render_b(oracle_mod.camera_from_host(cam), flat, oracle_mod.make_params(w, h, spp))
    assert np.array_equal(fix, fb) and st["rays_traced"] == stb["rays_traced"]
    if case.startswith("inside"):
        assert st["rays_traced"] > 3 * st["samples"]               # nothing escapes to the sky on the first ray
    sa, sta = oracle_mod.render_a(oracle_mod.camera_from_host(cam), flat, oracle_mod.make_params(w, h, spp))
    assert sta["rays_traced"] == st["rays_traced"]                  # and the literal (recursive) oracle walks the same paths


def test_resume_from_a_checkpoint_equals_one_oracle_render(tmp_path, renderer, oracle_mod, book1_flat):
    """SURVEY 8(f3), the whole flow: render 30 samples, save the exact sums (rtiow_amd.save_checkpoint), load them in a
    'later session', render samples 30..99 and add: the frame equals ONE Oracle-B render of 100 samples (main.rs:130-137 is
    the loop the two sessions split), and so do the RGBA8 bytes."""
    w, h, seed = 150, 84, 0x5EED5EED5EED
    cam = rt.book1_camera(w, h)
    renderer.upload_scene(book1_flat)
    _, first, _ = renderer.render(cam, rt.make_params(w, h, 30, seed=seed))
    path = str(tmp_path / "frame.ckpt.npz")
    rt.save_checkpoint(path, first, 30, seed)
    fix, done, seed2 = rt.load_checkpoint(path)
    assert done == 30 and seed2 == seed
    _, more, _ = renderer.render(cam, rt.make_params(w, h, 100 - done, sample_begin=done, seed=seed2))
    total = fix + more
    fb, _, _ = oracle_mod.render_b(oracle_mod.camera_from_host(cam), book1_flat, oracle_mod.make_params(w, h, 100, seed=seed))
    assert np.array_equal(total, fb)
    assert np.array_equal(renderer.resolve_rgba8(total, 100), oracle_mod.resolve_b(fb, 100))
