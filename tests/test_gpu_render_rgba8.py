"""rt_render_rgba8: main.rs:122-145 in ONE call through the C ABI -- render, Color::to_rgba (vec3.rs:403-421) and the row
flip (main.rs:141-145) with the exact sums kept on the device; only the RGBA8 bytes come back.  Bar: byte-identical to the
oracle's resolve of Oracle B's sums, and to the two-call path (rt_render -> host -> rt_resolve_rgba8) it replaces."""
import os
import subprocess

import numpy as np
import pytest

import rtiow_amd as rt

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("w,h,spp,flip", [(96, 54, 4, True), (96, 54, 4, False), (400, 225, 10, True), (37, 23, 41, True)])
def test_one_call_bytes_equal_the_oracles(renderer, oracle_mod, book1_flat, w, h, spp, flip):
    """(400,225,10) is BASELINE.json configs[0]; 41 spp goes through the LDS block sums, 4 and 10 straight to the frame buffer."""
    renderer.upload_scene(book1_flat)
    cam = rt.book1_camera(w, h)
    rgba, st = renderer.render_rgba8(cam, rt.make_params(w, h, spp, seed=7), flip=flip)
    fb, _, stb = oracle_mod.render_b(oracle_mod.camera_from_host(cam), book1_flat, oracle_mod.make_params(w, h, spp, seed=7))
    assert rgba.shape == (h, w, 4) and np.array_equal(rgba, oracle_mod.resolve_b(fb, spp, flip=flip))
    assert st["rays_traced"] == stb["rays_traced"] and st["samples"] == w * h * spp
    # ... and the path it replaces gives the same bytes
    _, fix, _ = renderer.render(cam, rt.make_params(w, h, spp, seed=7))
    assert np.array_equal(renderer.resolve_rgba8(fix, spp, flip=flip), rgba)


def test_one_call_on_a_shard(renderer, oracle_mod, book1_flat):
    """A shard's compact rows (tile_rows 2, shard 1 of 3); ACCUMULATE is ignored (host forms start from zero)."""
    w, h, spp = 64, 37, 3
    renderer.upload_scene(book1_flat)
    cam = rt.book1_camera(w, h)
    p = rt.make_params(w, h, spp, seed=3, tile_rows=2, shard_index=1, shard_count=3, flags=rt.RT_FLAG_ACCUMULATE)
    rgba, _ = renderer.render_rgba8(cam, p, flip=False)
    rows_j = rt.shard_row_indices(p)
    op = oracle_mod.make_params(w, h, spp, seed=3)
    fb, _, _ = oracle_mod.render_b(oracle_mod.camera_from_host(cam), book1_flat, op)
    want = oracle_mod.resolve_b(fb, spp, flip=False)[rows_j]
    assert np.array_equal(rgba, want)
    again, _ = renderer.render_rgba8(cam, p, flip=False)            # (a second call does not accumulate)
    assert np.array_equal(again, want)


def test_one_call_errors(renderer, book1_flat):
    renderer.upload_scene(book1_flat)
    cam = rt.book1_camera(16, 9)
    with pytest.raises(rt.RtiowHipError, match="spp >= 1"):
        renderer.render_rgba8(cam, rt.make_params(16, 9, 0))
    from rtiow_amd import _ffi
    import ctypes as C
    p = rt.make_params(16, 9, 1)
    rc = cam.to_rt_camera()
    assert _ffi.load().rt_render_rgba8(renderer._h, C.byref(rc), C.byref(p), 1, None, None) == -1
    assert "out_rgba" in _ffi.load().rt_last_error().decode()


def test_cpp_cli_one_call_and_two_calls_write_the_same_image(tmp_path):
    """host/rtiow_render: the single-device path is rt_render_rgba8; --two-calls takes the sums through host memory."""
    exe = os.path.join(ROOT, "host", "rtiow_render")
    if not os.path.exists(exe):
        pytest.skip("host CLI not built")
    a, b = str(tmp_path / "a.ppm"), str(tmp_path / "b.ppm")
    base = [exe, "--width", "120", "--height", "67", "--spp", "5", "--seed", "9"]
    subprocess.run(base + ["--out", a], check=True, timeout=300)
    subprocess.run(base + ["--out", b, "--two-calls"], check=True, timeout=300)
    assert open(a, "rb").read() == open(b, "rb").read()
