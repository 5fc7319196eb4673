"""bench.py's host-side logic (no GPU): the self-launch command for --gpus N, and the replayed-counter file."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_self_launch_command_is_a_torchrun_child_on_loopback():
    cmd = bench.launch_command(8, 29555, ["--gpus", "8", "--steps", "20", "--warmup", "5"])
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29555"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]


def test_kernel_name_carries_all_four_template_arguments():
    """roofline.kernel must join with rocprofv3's kernel trace by name: render_kernel<MODE, DIAG, SMALLGRID, U53, ITEMS>."""
    assert bench.kernel_name(5, 1) == "rt::render_kernel<5, false, true, false, 256>"       # the shipped small-grid kernel
    assert bench.kernel_name(5, 5) == "rt::render_kernel<5, false, true, false, 1024>"      # ... on a launch of >= 2 x 10^8 pixel-samples (the bench default)
    assert bench.kernel_name(5, 4) == "rt::render_kernel<5, false, false, false, 1024>"     # large grids (10k spheres, 1920x1080x256)
    assert bench.kernel_name(5, 3) == "rt::render_kernel<5, false, true, true, 256>"        # RT_FLAG_UNIFORM53
    assert bench.kernel_name(0, 2) == "rt::render_kernel<0, false, false, true, 256>"


def test_a_rank_that_never_arrives_ends_the_run_within_the_timeout():
    """bench.py gives init_process_group an explicit timeout (RTIOW_DIST_TIMEOUT_S, default 120 s): with WORLD_SIZE=2 and
    only rank 0 started, the rendezvous must give up non-zero after that time, not after torch's 10-30 minute defaults.
    (gloo rehearsal path; the rendezvous comes before anything touches a GPU, so this runs on any box.)"""
    import socket
    import time
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RTIOW_DIST_TIMEOUT_S="5")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--steps", "1",
                        "--warmup", "0"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and time.time() - t0 < 90, (r.returncode, r.stderr[-500:])
    assert "imeout" in r.stderr or "timed out" in r.stderr, r.stderr[-800:]


def test_help_needs_neither_torch_nor_a_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "--rehearse-on-one-gpu" in r.stdout


def test_pmc_replay_entries_name_their_source_and_kernel():
    path = os.path.join(ROOT, "profiles", "pmc_replay.json")
    entries = json.load(open(path))["entries"]
    assert any(e["config"] == [1200, 675, 500, 1] for e in entries)      # the bench default
    for e in entries:
        assert "NOT measured in this bench run" in e["source"] and e["source"].startswith("profiles/")
        assert len(e["kernel_source_sha"]) == 16 and e["valu_insts_per_launch"] > 0 and 0 < e["valu_busy"] <= 1.5
        assert os.path.exists(os.path.join(ROOT, e["source"].split(":")[0]))
    sha = bench.kernel_source_sha()
    if not any(e["kernel_source_sha"] == sha for e in entries):
        print(f"note: profiles/pmc_replay.json was taken on other kernel sources than {sha}: bench.py will null the counters")


def test_recorded_bench_line_has_the_contracts_fields():
    """The line recorded in profiles/ (this round's last `python bench.py --steps 20 --warmup 5` on an MI355X) carries every
    field the driver and the judge read, and its roofline numbers are consistent with each other."""
    import json
    path = os.path.join(ROOT, "profiles", "r05_bench_n1.json")
    d = json.loads(open(path).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "rmse_vs_cpu", "parity_vs_cpu"):
        assert k in d, k
    # the metric is BASELINE.json's, both halves: throughput and the per-pixel RMSE vs the CPU render of the same rows
    assert d["metric"] == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    assert d["rmse_vs_cpu"] < 1e-4 and d["parity_vs_cpu"]["rows"] >= 100 and d["parity_vs_cpu"]["rgba8_rows_identical"] is True
    assert d["parity_vs_cpu"]["samples"] == d["parity_vs_cpu"]["rows"] * 1200 * 500
    assert d["unit"] == "Msamples/s" and d["n_gpus"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    cfg = d["config"]
    assert (cfg["width"], cfg["height"], cfg["spp"]) == (1200, 675, 500)
    # value = samples of K steps / wall time
    assert abs(d["value"] - cfg["width"] * cfg["height"] * cfg["spp"] / d["ms_per_step"] / 1e3) < 0.01 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms", "counters_source"):
        assert k in r, k
    assert 0.0 < r["frac"] <= 1.0 and abs(r["achieved"] / r["peak"] - r["frac"]) < 0.01
    # the peak is a hardware figure, not achieved / utilisation: 1024 SIMDs x 2.4 GHz / (2..4 issue cycles per instruction)
    assert 1024 * 2.4 / 4.0 < r["peak"] <= r["peak_uniform"] == 1228.8 and 0.5 < r["valu_busy"] <= 1.0
    assert d["config"]["library_source_sha"] == cfg["kernel_source_sha"]
    assert abs(r["kernel_ms"] - r["kernel_ms_rocprof_avg"]) < 0.03 * r["kernel_ms"]         # HIP events vs rocprofv3 --stats
    assert r["kernel_ms"] <= d["ms_per_step"]
    # SURVEY 8(d)'s separate end-to-end figure: the host-buffer calls of main.rs:122-145, never `value`; the one-call form costs < 0.6 ms over its kernel
    e = d["end_to_end"]
    for which in ("headline", "configs[1]"):
        assert e[which]["same_bytes"] is True and 0.0 < e[which]["one_call"]["overhead_ms"] <= 0.6
        assert e[which]["one_call"]["ms"] <= e[which]["two_calls"]["ms"] and e[which]["one_call"]["ms"] >= e[which]["one_call"]["kernel_ms"]
    assert (e["headline"]["width"], e["headline"]["height"], e["headline"]["spp"]) == (1200, 675, 500) and e["configs[1]"]["spp"] == 100
    assert any("UNIFORM53" in o.get("config", "") for o in d["other_configs"])          # the price of 53-bit draws, in the driver's own line
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["unit"] == d["unit"]
    # (whether the recorded line belongs to the kernel sources of this tree is reported, not enforced: bench.py itself
    #  nulls the replayed counters when the sources have moved on)
    if cfg["kernel_source_sha"] != bench.kernel_source_sha():
        import pytest
        pytest.skip("profiles/r05_bench_n1.json was recorded on other kernel sources (%s)" % cfg["kernel_source_sha"])


def test_rmse_helper_on_the_two_oracles():
    """bench.rmse_vs_cpu (the second half of BASELINE.json's metric) on CPU data: Oracle B's exact sums stand in for the GPU's
    (they are what the GPU must reproduce bit for bit), Oracle A is the CPU render: RMSE of the linear means ~1e-10, RGBA8
    bytes identical; and the helper notices a frame that is off by one sample in one pixel."""
    import numpy as np
    import oracle
    import rtiow_amd as rt
    flat = rt.random_scene(1).flatten()
    w, h, spp = 64, 36, 8
    cam = oracle.book1_camera(w, h)
    p = oracle.make_params(w, h, spp, rows=(0, h, 3))
    fix, _, _ = oracle.render_b(cam, flat, p)
    sums, _ = oracle.render_a(cam, flat, p)
    rgba = oracle.resolve_b(fix, spp, flip=False)
    par = bench.rmse_vs_cpu(fix, rgba, sums, spp)
    assert par["rmse"] < 1e-9 and par["rgba8_rows_identical"] is True and par["rgba8_bytes_differing"] == 0
    bad = fix.copy()
    bad[2, 5, 1] += np.uint64(1 << 32)                                   # one more unit of radiance in one channel of one pixel
    par = bench.rmse_vs_cpu(bad, rgba, sums, spp)
    assert par["max_abs_diff"] > 0.12 and par["rmse"] > 1e-4
    cb, rows_j, cpu_sums = bench.cpu_baseline(flat, w, h, spp, target_seconds=0.2, min_rows=10)
    assert len(rows_j) == cpu_sums.shape[0] >= 10 and cb["kind"] == "port" and cb["value"] > 0
    assert np.array_equal(rows_j, np.arange(0, h, rows_j[1] - rows_j[0])[:len(rows_j)])


def test_weak_efficiency_is_formed_against_the_committed_single_gpu_line():
    """At N > 1 the line carries T(--weak-baseline) / T(this run) at equal samples per GPU, labelled cross-run;
    no value for a rehearsal on one GPU or for another per-GPU sample count."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench_weak_baseline.json")))
    base = json.loads(open(files[-1]).read().strip().splitlines()[-1])
    spg = base["config"]["samples_per_gpu"]
    assert spg == 3840 * 2160 * 500 == 7680 * 4320 * 125                    # configs[2] on one GPU = an eighth of configs[4]
    e = bench.weak_efficiency(base["ms_per_step"] / 0.95, spg, bench.kernel_source_sha(), False)
    assert abs(e["value"] - 0.95) < 1e-3 and "cross-run" in e["kind"] and e["baseline_source"].startswith("profiles/")
    assert bench.weak_efficiency(600.0, spg, "x", True)["value"] is None      # rehearsal
    assert bench.weak_efficiency(600.0, spg // 2, "x", False)["value"] is None
