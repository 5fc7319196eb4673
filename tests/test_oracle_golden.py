"""Oracle A vs Oracle B, and both against the committed golden fixture."""
import numpy as np

W, H, SPP = 32, 18, 4


def test_oracle_b_reproduces_golden(oracle_mod, book1_flat, golden_small):
    cam = oracle_mod.book1_camera(W, H)
    fix, sm, st = oracle_mod.render_b(cam, book1_flat, oracle_mod.make_params(W, H, SPP, seed=1))
    assert np.array_equal(fix, golden_small["fix"])
    assert np.array_equal(sm, golden_small["sum_f32"])
    assert st["rays_traced"] == int(golden_small["rays_b"])
    assert np.array_equal(oracle_mod.resolve_b(fix, SPP), golden_small["rgba"])
    assert st["depth_hist"] == golden_small["depth_hist"].tolist()


def test_oracle_a_reproduces_golden(oracle_mod, book1_flat, golden_small):
    cam = oracle_mod.book1_camera(W, H)
    sa, st = oracle_mod.render_a(cam, book1_flat, oracle_mod.make_params(W, H, SPP, seed=1))
    assert np.array_equal(sa, golden_small["sum_a_f64"])
    assert st["rays_traced"] == int(golden_small["rays_a"])


def test_contract_b_is_a_up_to_truncation(oracle_mod, book1_flat):
    """B = A + iterative throughput + 2^-32 truncation: same paths (ray counts), sums within
    spp quanta (each sample truncates by < 1 quantum; the product order moves ~1e-16)."""
    w, h, spp = 96, 54, 8
    cam = oracle_mod.book1_camera(w, h)
    p = oracle_mod.make_params(w, h, spp, seed=7)
    sa, sta = oracle_mod.render_a(cam, book1_flat, p)
    fix, _, stb = oracle_mod.render_b(cam, book1_flat, p)
    assert sta["rays_traced"] == stb["rays_traced"]
    assert sta["depth_hist"] == stb["depth_hist"]
    diff_quanta = np.abs(sa * 2.0 ** 32 - fix.astype(np.float64))
    assert diff_quanta.max() <= spp
    rmse = np.sqrt(np.mean((sa / spp - fix.astype(np.float64) / 2.0 ** 32 / spp) ** 2))
    assert rmse < 1e-9                                   # north_star gate is 1e-4
    assert np.array_equal(oracle_mod.resolve_a(sa, spp), oracle_mod.resolve_b(fix, spp))


def test_thread_count_and_row_subsets_do_not_change_results(oracle_mod, book1_flat):
    cam = oracle_mod.book1_camera(W, H)
    full, _, _ = oracle_mod.render_b(cam, book1_flat, oracle_mod.make_params(W, H, SPP, seed=1, nthreads=3))
    one, _, _ = oracle_mod.render_b(cam, book1_flat, oracle_mod.make_params(W, H, SPP, seed=1, nthreads=1))
    assert np.array_equal(full, one)
    sub, _, _ = oracle_mod.render_b(cam, book1_flat, oracle_mod.make_params(W, H, SPP, seed=1, rows=(3, 15, 4)))
    assert np.array_equal(sub, full[3:15:4])


def test_sample_ranges_are_additive(oracle_mod, book1_flat):
    cam = oracle_mod.book1_camera(W, H)
    a, _, _ = oracle_mod.render_b(cam, book1_flat, oracle_mod.make_params(W, H, 3, seed=1))
    b, _, _ = oracle_mod.render_b(cam, book1_flat, oracle_mod.make_params(W, H, 1, sample_begin=3, seed=1))
    full, _, _ = oracle_mod.render_b(cam, book1_flat, oracle_mod.make_params(W, H, 4, seed=1))
    assert np.array_equal(a + b, full)


def test_statistics_of_the_book1_scene(oracle_mod, book1_flat):
    """BASELINE.md section 2 workload figures (rays/sample, termination mix)."""
    w, h, spp = 160, 90, 16
    cam = oracle_mod.book1_camera(w, h)
    _, st = oracle_mod.render_a(cam, book1_flat, oracle_mod.make_params(w, h, spp, seed=3))
    rps = st["rays_traced"] / st["samples"]
    assert 2.4 < rps < 2.9
    assert st["end_sky"] / st["samples"] > 0.99
    assert st["end_sky"] + st["end_absorb"] + st["end_depth"] == st["samples"]
