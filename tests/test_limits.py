"""The two limits of the boundary that differ from the reference's unbounded types (include/rtiow_hip.h):

  * the scene list (shapes/mod.rs:52 is a Vec of any length): RT_MAX_SPHERES = 2^24 since round 4 (rounds 1-3: 65 535) -- a
    70 227-sphere scene against Oracle B;
  * the pixel sum (main.rs:127,135 is an f64): contract C5 clamps a sample's channel at RT_SAMPLE_CLAMP = 2^16 and sums
    u64 -- a scene with albedos of 3 (samples of up to 3^50) against the LITERAL Oracle A, whose sums are the reference's
    unbounded f64: the RGBA8 bytes must be the same, saturated pixels included (with the earlier clamp at 2^30 four saturated
    samples wrapped the sum).
"""
import os

import numpy as np
import pytest

import rtiow_amd as rt
from rtiow_amd.scene import RT_DIALECTRIC

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bright_scene(book1_flat, factor=3.0):
    flat = book1_flat.copy()
    opaque = flat["kind"] != RT_DIALECTRIC
    flat["albedo"][opaque] *= factor
    return flat


def test_header_states_both_limits():
    hdr = open(os.path.join(ROOT, "include", "rtiow_hip.h")).read()
    assert "#define RT_MAX_SPHERES (1 << 24)" in hdr and "#define RT_SAMPLE_CLAMP 65536.0" in hdr
    assert "CANNOT wrap while a pixel has received FEWER THAN 65 536 samples" in hdr


def test_c5_range_is_65535_saturated_samples(oracle_mod):
    """The largest quantised sample is exactly 2^48: 65 535 of them fit a u64, 65 536 sum to 2^64."""
    q = int(oracle_mod.load().oracle_b_quantize(float("inf")))
    assert q == int(oracle_mod.load().oracle_b_quantize(65536.0)) == 1 << 48
    assert 65535 * q < 2 ** 64 <= 65536 * q


def test_c5_clamp_gives_the_references_bytes_for_albedos_above_one(oracle_mod, book1_flat):
    """CPU: Oracle B (clamped u64 sums, the kernel's contract) against Oracle A (the reference's unbounded f64 sums) on a scene
    whose diffuse and metal albedos are tripled: 3^k grows past 2^16 within 11 bounces, so many samples are clamped and many
    pixels saturate.  to_rgba must give the same bytes: a clamped sample alone makes the pixel's mean >= 1 -> 255 on both sides."""
    flat = bright_scene(book1_flat)
    w, h, spp = 96, 54, 16
    cam = oracle_mod.book1_camera(w, h)
    p = oracle_mod.make_params(w, h, spp)
    fix, _, _ = oracle_mod.render_b(cam, flat, p)
    sums, _ = oracle_mod.render_a(cam, flat, p)
    clamped_samples_possible = (sums > 65536.0).any()
    assert clamped_samples_possible                                        # the scene does reach the clamp
    assert int(fix.max()) <= spp << 48                                     # no sum can pass spp * 2^16 (in units of 2^-32): far from 2^64
    assert (sums.max() > 2.0 ** 32)                                        # ... where the unbounded sum is beyond what a u64 of 2^-32 quanta holds
    a, b = oracle_mod.resolve_a(sums, spp), oracle_mod.resolve_b(fix, spp)
    assert np.array_equal(a, b)
    assert (a[..., :3] == 255).mean() > 0.2                                # and a good part of the frame is saturated


@pytest.mark.gpu
def test_gpu_albedos_above_one_equal_both_oracles(renderer, oracle_mod, book1_flat):
    flat = bright_scene(book1_flat)
    w, h, spp = 160, 90, 40                                                # (40 spp: the LDS block-sum path)
    renderer.upload_scene(flat)
    cam = rt.book1_camera(w, h)
    _, fix, st = renderer.render(cam, rt.make_params(w, h, spp))
    ocam, op = oracle_mod.camera_from_host(cam), oracle_mod.make_params(w, h, spp)
    fb, _, stb = oracle_mod.render_b(ocam, flat, op)
    assert np.array_equal(fix, fb) and st["rays_traced"] == stb["rays_traced"]
    sums, _ = oracle_mod.render_a(ocam, flat, op)
    assert np.array_equal(renderer.resolve_rgba8(fix, spp), oracle_mod.resolve_a(sums, spp))
    renderer.upload_scene(book1_flat)


@pytest.mark.gpu
def test_seventy_thousand_spheres_bit_exact(renderer, oracle_mod):
    """More spheres than 16-bit candidate numbers could name (rounds 1-3 rejected n > 65 535): a 265 x 265 lattice of small
    spheres + ground + the three big ones, against Oracle B's ordered scan of the whole list; columns of the filter table
    beyond 65 535 are in use (the grid has 63 x 63 cells)."""
    flat = rt.random_scene(1, grid=(-132, 132)).flatten()
    assert len(flat) > 70000
    dims, grid, slot_of = rt.tile_layout_host(flat)
    assert dims[0] > 42 and (np.nonzero(slot_of >= 0)[0].max() > 65535)
    renderer.upload_scene(flat)
    w, h, spp = 64, 36, 2
    cam = rt.book1_camera(w, h)
    _, fix, st = renderer.render(cam, rt.make_params(w, h, spp))
    fb, _, stb = oracle_mod.render_b(oracle_mod.camera_from_host(cam), flat, oracle_mod.make_params(w, h, spp, nthreads=8))
    assert np.array_equal(fix, fb) and st["rays_traced"] == stb["rays_traced"] and st["n_spheres"] == len(flat)
    # a camera low over the lattice, looking along it: long in-slab rays over many cells
    low = rt.Camera((120.0, 0.6, 118.0), (-100.0, 0.2, -90.0), (0.0, 1.0, 0.0), 35.0, w / h, 0.02, 40.0)
    _, fix, st = renderer.render(low, rt.make_params(w, h, spp, seed=5))
    fb, _, stb = oracle_mod.render_b(oracle_mod.camera_from_host(low), flat, oracle_mod.make_params(w, h, spp, seed=5, nthreads=8))
    assert np.array_equal(fix, fb) and st["rays_traced"] == stb["rays_traced"]
    # WITHOUT the grid (what a scene of more than ~130 000 filtered spheres gets: no G x G of <= 63 x 63 cells holds it): every wave
    # scans every tile of the table -- the list_all path with tile numbers >= 2 048 and pool columns >= 2^16
    os.environ["RTIOW_NO_GRID"] = "1"
    try:
        rn = rt.Renderer(0)
        rn.upload_scene(flat)
        _, fix_n, st_n = rn.render(low, rt.make_params(w, h, spp, seed=5))
        rn.close()
    finally:
        os.environ.pop("RTIOW_NO_GRID")
    assert np.array_equal(fix_n, fb) and st_n["rays_traced"] == stb["rays_traced"]
    # the VALU cross-check filter keeps 16-bit candidate lists: it must refuse this scene, not truncate it
    os.environ["RTIOW_SCAN_MODE"] = "1"
    try:
        r1 = rt.Renderer(0)
        with pytest.raises(rt.RtiowHipError, match="at most 65535 spheres"):
            r1.upload_scene(flat)
        r1.close()
    finally:
        os.environ.pop("RTIOW_SCAN_MODE")
