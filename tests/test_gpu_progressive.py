"""Progressive passes that OVERLAP (SURVEY.md 8(f3) meets the end-of-launch tail): main.rs:130-137's sample loop split over launches
with sample_begin + RT_FLAG_ACCUMULATE, pass k + 1 issued on another stream than pass k so that it fills pass k's tail.  A context
holds the per-launch state (work counter, statistics, events) of two launches.  The sums are exact integers added with atomics:
the frame must equal ONE Oracle-B render of all the samples, bit for bit, however the passes interleave."""
import numpy as np
import pytest
import torch

import rtiow_amd as rt

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("w,h,spp_pass,passes,overlapped", [(160, 90, 40, 5, 0), (96, 54, 7, 3, 0), (64, 36, 41, 4, 1), (112, 63, 80, 3, 1),
                                                            (80, 45, 150, 2, 1)])
def test_overlapping_passes_on_two_streams_equal_one_oracle_render(renderer, oracle_mod, book1_flat, w, h, spp_pass, passes, overlapped):
    """40/41 spp per pass: block sums in LDS; 7: straight to the frame buffer.  Four and five passes: every slot is reused
    (the third launch makes its stream wait for the first).  RT_FLAG_OVERLAPPED (the caller says that the passes overlap): work blocks
    of 1 024 pixel-samples whatever the launch's size, from 69 samples per pixel on (rt_stats.kernel_variant bit 2) -- the same frame."""
    renderer.upload_scene(book1_flat)
    cam = rt.book1_camera(w, h)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    d_fix = torch.zeros((h, w, 3), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()                                    # the buffer is zero before any stream adds to it
    for k in range(passes):
        p = rt.make_params(w, h, spp_pass, sample_begin=k * spp_pass, seed=5,
                           flags=rt.RT_FLAG_ACCUMULATE | (rt.RT_FLAG_OVERLAPPED if overlapped else 0))
        renderer.render_device(cam, p, d_fix.data_ptr(), streams[k & 1].cuda_stream)
    torch.cuda.synchronize()
    st = renderer.last_stats()
    assert st["samples"] == w * h * spp_pass                    # (the latest launch)
    if st["scan_mode"] == 5:
        assert bool(st["kernel_variant"] & 4) == bool(overlapped and spp_pass >= 69)
    got = d_fix.cpu().numpy().view(np.uint64)
    want, _, _ = oracle_mod.render_b(oracle_mod.camera_from_host(cam), book1_flat, oracle_mod.make_params(w, h, spp_pass * passes, seed=5))
    assert np.array_equal(got, want)


def test_two_launches_in_flight_keep_their_own_counters(renderer, oracle_mod, book1_flat):
    """Two DIFFERENT frames at once from one context (two streams, two buffers): each equals its own oracle render -- the
    work counters and statistics of the two launches do not mix."""
    renderer.upload_scene(book1_flat)
    a_cam, b_cam = rt.book1_camera(200, 112), rt.book1_camera(120, 67)
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    fa = torch.zeros((112, 200, 3), dtype=torch.int64, device="cuda")
    fb = torch.zeros((67, 120, 3), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    renderer.render_device(a_cam, rt.make_params(200, 112, 48, seed=1), fa.data_ptr(), s0.cuda_stream)
    renderer.render_device(b_cam, rt.make_params(120, 67, 5, seed=2), fb.data_ptr(), s1.cuda_stream)
    st_b = renderer.last_stats()
    torch.cuda.synchronize()
    wa, _, _ = oracle_mod.render_b(oracle_mod.camera_from_host(a_cam), book1_flat, oracle_mod.make_params(200, 112, 48, seed=1))
    wb, _, stb = oracle_mod.render_b(oracle_mod.camera_from_host(b_cam), book1_flat, oracle_mod.make_params(120, 67, 5, seed=2))
    assert np.array_equal(fa.cpu().numpy().view(np.uint64), wa)
    assert np.array_equal(fb.cpu().numpy().view(np.uint64), wb)
    assert st_b["samples"] == 120 * 67 * 5 and st_b["rays_traced"] == stb["rays_traced"]


def test_cpp_cli_overlapping_passes_write_the_single_calls_image(tmp_path):
    """host/rtiow_render --passes N: the compiled host issues N additive launches alternately on two streams of one context
    (rt_render_device + RT_FLAG_ACCUMULATE | RT_FLAG_OVERLAPPED, then rt_resolve_rgba8_device); the image file is the single call's, byte for byte."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "host", "rtiow_render")
    if not os.path.exists(exe):
        pytest.skip("host CLI not built")
    one, many, odd = (str(tmp_path / n) for n in ("one.ppm", "many.ppm", "odd.ppm"))
    base = [exe, "--width", "160", "--height", "90", "--spp", "123", "--seed", "4"]
    subprocess.run(base + ["--out", one], check=True, timeout=300)
    subprocess.run(base + ["--out", many, "--passes", "3"], check=True, timeout=300)       # 41 samples per pass: LDS block sums
    subprocess.run(base + ["--out", odd, "--passes", "7"], check=True, timeout=300)        # 18 / 17 samples per pass: direct adds, uneven passes
    ref = open(one, "rb").read()
    assert open(many, "rb").read() == ref and open(odd, "rb").read() == ref
