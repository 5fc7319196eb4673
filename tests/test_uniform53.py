"""RT_FLAG_UNIFORM53: every uniform from TWO consecutive Philox words (53 random bits, what rand 0.8.5's gen::<f64>() draws
at main.rs:131-132, materials.rs:96) instead of one word's 32 bits; same draw order.

CPU: the oracle's word -> uniform rule against Philox words computed independently; the two streams render the same image
statistically.  GPU: the kernel's 53-bit instantiations against Oracle B, bit for bit (the kernel and the oracle implement the
rule, the runs of consecutive words and the f64 rejection tests independently)."""
import numpy as np
import pytest

import rtiow_amd as rt


def test_oracle_uniforms_are_the_stated_function_of_the_philox_words(oracle_mod):
    seed, pixel, sample = 0x1234567887654321, 777, 5
    key = (seed & 0xFFFFFFFF, seed >> 32)
    words = []
    for e in range(4):
        words += list(oracle_mod.philox((pixel, sample, e, 0), key))
    u32 = oracle_mod.uniforms(seed, pixel, sample, 16)
    assert np.array_equal(u32, np.array([w / 2.0 ** 32 for w in words]))          # the default: all 32 bits of one word
    u53 = oracle_mod.uniforms(seed, pixel, sample, 8, uniform53=True)
    want = np.array([(((words[2 * k] << 32) | words[2 * k + 1]) >> 11) / 2.0 ** 53 for k in range(8)])
    assert np.array_equal(u53, want)
    assert ((u53 >= 0) & (u53 < 1)).all() and len(set(u53)) == 8
    # a 53-bit uniform refines the 32-bit one of its first word: same leading bits
    assert np.all(np.floor(u53 * 2.0 ** 32) == np.array([words[2 * k] for k in range(8)]))


def test_oracle_53_bit_stream_renders_the_same_image_statistically(oracle_mod, book1_flat):
    w, h, spp = 96, 54, 64
    cam = oracle_mod.book1_camera(w, h)
    a, _, sa = oracle_mod.render_b(cam, book1_flat, oracle_mod.make_params(w, h, spp))
    b, _, sb = oracle_mod.render_b(cam, book1_flat, oracle_mod.make_params(w, h, spp, uniform53=True))
    ma, mb = a.astype(np.float64) / 2.0 ** 32 / spp, b.astype(np.float64) / 2.0 ** 32 / spp
    assert not np.array_equal(a, b)                                      # another stream
    assert abs(ma.mean() - mb.mean()) < 0.004                            # the same estimator (MC noise of the mean ~1e-3)
    assert abs(sa["rays_traced"] / sa["samples"] - sb["rays_traced"] / sb["samples"]) < 0.02
    # literal recursion (Oracle A) and the kernel contract (Oracle B) agree under the flag exactly as they do without it
    sa2, st2 = oracle_mod.render_a(cam, book1_flat, oracle_mod.make_params(w, h, 4, uniform53=True))
    fb2, _, stb2 = oracle_mod.render_b(cam, book1_flat, oracle_mod.make_params(w, h, 4, uniform53=True))
    assert st2["rays_traced"] == stb2["rays_traced"]
    assert np.abs(fb2.astype(np.float64) / 2.0 ** 32 - sa2).max() < 4 * 2.0 ** -32 * 4 + 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,spp,begin,flags", [(160, 90, 8, 0, 0), (120, 68, 40, 3, 0), (64, 36, 5, 0, rt.RT_FLAG_NO_FILTER)])
def test_gpu_53_bit_uniforms_bit_exact_vs_oracle(renderer, oracle_mod, book1_flat, w, h, spp, begin, flags):
    renderer.upload_scene(book1_flat)
    cam = rt.book1_camera(w, h)
    sm, fix, st = renderer.render(cam, rt.make_params(w, h, spp, sample_begin=begin, flags=flags | rt.RT_FLAG_UNIFORM53))
    fb, sb, stb = oracle_mod.render_b(oracle_mod.camera_from_host(cam), book1_flat,
                                      oracle_mod.make_params(w, h, spp, sample_begin=begin, uniform53=True))
    assert np.array_equal(fix, fb) and np.array_equal(sm, sb)
    assert st["rays_traced"] == stb["rays_traced"]
    assert st["kernel_variant"] & 2
    _, fix24, _ = renderer.render(cam, rt.make_params(w, h, spp, sample_begin=begin, flags=flags))
    assert not np.array_equal(fix24, fix)                                # the default stream is another one


@pytest.mark.gpu
def test_gpu_53_bit_uniforms_on_the_large_grid_kernel_and_hand_materials(renderer, oracle_mod):
    """The general (large-grid) instantiation on a 3 000-sphere scene, and a scene of glass and fuzzy metal only (every bounce
    draws: the Dialectric's single 53-bit draw, Metal's unit-sphere tries)."""
    mid = rt.random_scene(1, grid=(-27, 27)).flatten()
    w, h, spp = 96, 54, 6
    cam = rt.book1_camera(w, h)
    renderer.upload_scene(mid)
    _, fix, st = renderer.render(cam, rt.make_params(w, h, spp, flags=rt.RT_FLAG_UNIFORM53))
    fb, _, stb = oracle_mod.render_b(oracle_mod.camera_from_host(cam), mid, oracle_mod.make_params(w, h, spp, uniform53=True))
    assert np.array_equal(fix, fb) and st["rays_traced"] == stb["rays_traced"] and st["kernel_variant"] == 2
    world = rt.HittableList()
    world.push(rt.Sphere(rt.Point3(0, -1000, 0), 1000, rt.Metal(rt.Color(0.8, 0.8, 0.8), 0.6)))
    for k in range(-3, 4):
        world.push(rt.Sphere(rt.Point3(2.0 * k, 1, 0.5 * k), 1.0, rt.Dialectric(1.5) if k % 2 else rt.Metal(rt.Color(0.7, 0.6, 0.5), 1.0)))
    flat = world.flatten()
    renderer.upload_scene(flat)
    _, fix, st = renderer.render(cam, rt.make_params(w, h, 12, flags=rt.RT_FLAG_UNIFORM53))
    fb, _, stb = oracle_mod.render_b(oracle_mod.camera_from_host(cam), flat, oracle_mod.make_params(w, h, 12, uniform53=True))
    assert np.array_equal(fix, fb) and st["rays_traced"] == stb["rays_traced"]


@pytest.mark.gpu
def test_gpu_53_bit_uniforms_refuse_the_combinations_that_are_not_built(renderer, book1_flat):
    renderer.upload_scene(book1_flat)
    with pytest.raises(rt.RtiowHipError, match="RT_FLAG_UNIFORM53"):
        renderer.render(rt.book1_camera(16, 9), rt.make_params(16, 9, 1, flags=rt.RT_FLAG_UNIFORM53 | rt.RT_FLAG_DIAG_STATS))
