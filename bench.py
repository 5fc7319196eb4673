#!/usr/bin/env python3
"""bench.py -- Msamples/s (pixels x spp) of the render path on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

With --gpus N > 1 and no WORLD_SIZE in the environment this process starts the N ranks itself
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` as a
CHILD process, before anything here has touched the GPU) and relays rank 0's JSON line and the
child's exit code; started by an external torch.distributed.run it is one of those ranks.

Workloads (BASELINE.json):
  N = 1   book-1 final scene, 1200x675, 500 spp, depth 50 -- the configuration the north_star
          target (>= 1 Gsample/s) is quoted on.  configs[1] (100 spp), configs[2] (3840x2160x500)
          and configs[3] (10k spheres, 1920x1080x256) are timed after it and reported under
          "other_configs".
  N > 1   SURVEY.md 8(e)'s weak-scaling pair: configs[4]'s geometry (7680x4320) with 125 x N spp,
          rows dealt round-robin to the N ranks (N = 8 is configs[4] itself: 1000 spp); every GPU
          traces 4.147 G pixel-samples per step, exactly configs[2]'s count on one GPU.

One "step" = one complete frame: every rank renders its interleaved rows with inputs resident in
HBM, ONE gather (RCCL) brings the exact sums to rank 0, rank 0 resolves them to RGBA8
(Color::to_rgba + flip).  Prints ONE JSON line on rank 0.

The JSON also carries
  roofline      for the dominant kernel (render_kernel), measured live with the library's HIP
                events on the launch stream; counters that need rocprofv3 --pmc are replayed from
                profiles/ and labelled with their source (null when the kernel sources differ
                from the profiled ones);
  cpu_baseline  Oracle A (the literal f64 CPU restatement of the reference,
                oracle/oracle_f64.c) timed on this host's cores on a bounded row
                subset of the same frame (rank 0, N=1 only).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

PEAK_FP32_VECTOR_TFLOPS = 157.3        # MI355X_MICROARCH.md, "Peak FP32 (vector)"
PEAK_HBM_GBPS = 8000.0                 # MI355X_MICROARCH.md, HBM3E spec peak
PEAK_BF16_MFMA_TFLOPS = 2500.0         # MI355X_MICROARCH.md, dense bf16 matrix peak
FLOP_PER_TEST = 17                     # per ray-sphere test
FLOP_PER_RAY_FIXED = 65                # hit finalisation + shading per ray
N_SIMD = 1024                          # 256 CUs x 4 SIMDs
SPEC_CLOCK_GHZ = 2.4                   # MI355X_MICROARCH.md: peak engine clock
# hardware issue cost of one wave64 vector instruction on a SIMD-32 (MI355X_MICROARCH.md, "Per-instruction cycle
# constants"): 2 cycles for f32/int/bit ops, 4 for f64 add/mul/fma (78.6 TFLOP/s f64 vector peak = 16 lanes/clk/SIMD),
# 8 for the transcendental unit (v_rcp/v_rsq/v_sqrt_f32)
ISSUE_CYCLES = {"plain": 2.0, "f64": 4.0, "trans": 8.0}

KERNEL_SOURCES = ("rtiow_amd/csrc/rt_kernels.hpp", "rtiow_amd/csrc/rt_device.hpp", "rtiow_amd/csrc/rt_api.hip", "rtiow_amd/csrc/rt_diag.hpp")


def kernel_source_sha():
    """Identifies the kernel a profile was taken on: sha256 over the product's kernel sources (csrc/xcheck/ belongs to the cross-check
    build only)."""
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def host_cpu_share():
    """CPUs this process may actually use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def baseline_metric():
    """BASELINE.json's metric string (the bench measures THAT metric); the literal is the fallback on a box without the file."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "Msamples/sec (pixels\u00d7spp) on book-1 final scene; per-pixel RMSE vs CPU"


def cpu_baseline(flat, width, height, spp, target_seconds=12.0, min_rows=100):
    """Oracle A on a bounded, evenly strided row subset of the same frame (at least `min_rows` rows of it: the
    subset also carries the RMSE-vs-CPU half of the metric).  Returns (json object, rows j, f64 sums [rows, W, 3])."""
    import oracle
    cam = oracle.book1_camera(width, height)
    threads = host_cpu_share()
    probe_step = max(1, height // 8)
    p = oracle.make_params(width, height, max(1, spp // 10), rows=(0, height, probe_step), nthreads=threads)
    _, st = oracle.render_a(cam, flat, p)
    rate = st["samples"] / st["seconds"]
    rows_wanted = max(float(min(min_rows, height)), target_seconds * rate / (width * spp))
    step = int(min(height, max(1, height // int(min(height, round(rows_wanted))))))
    p = oracle.make_params(width, height, spp, rows=(0, height, step), nthreads=threads)
    sums, st = oracle.render_a(cam, flat, p)
    nrows = oracle.n_rows(p)
    obj = {
        "value": round(st["samples"] / st["seconds"] / 1e6, 4), "unit": "Msamples/s",
        "cores": int(st["threads_used"]), "kind": "port",
        "sample": f"Oracle A (literal f64 restatement, oracle/oracle_f64.c), rows j=0,{step},2*{step},... "
                  f"({nrows} of {height} rows) of the {width}x{height}x{spp}spp frame, {st['samples']} samples in "
                  f"{st['seconds']:.2f} s on {st['threads_used']} host threads",
    }
    return obj, np.arange(0, height, step, dtype=np.int64)[:nrows], sums


def rmse_vs_cpu(gpu_fix_rows, gpu_rgba_rows_bottom_up, cpu_sums, spp):
    """The second half of BASELINE.json's metric: per-pixel RMSE of the GPU frame against the CPU render of the same
    rows at matched seeds, on LINEAR radiance means (sqrt(mean over pixels and channels of
    (GPU_mean - CPU_mean)^2), before gamma) -- main.rs:135-137's pixel_color / spp on both sides -- and whether
    Color::to_rgba gives the same bytes.  gpu_fix_rows: u64 exact sums [rows, W, 3] (quantum 2^-32)."""
    import oracle
    gpu_mean = gpu_fix_rows.astype(np.float64) * (1.0 / 4294967296.0) / spp
    cpu_mean = cpu_sums / spp
    diff = gpu_mean - cpu_mean
    cpu_rgba = oracle.resolve_a(cpu_sums, spp, flip=False)
    return {
        "rmse": float(np.sqrt(np.mean(diff * diff))), "max_abs_diff": float(np.abs(diff).max()),
        "rgba8_rows_identical": bool(np.array_equal(cpu_rgba, gpu_rgba_rows_bottom_up)),
        "rgba8_bytes_differing": int((cpu_rgba != gpu_rgba_rows_bottom_up).sum()),
    }


def kernel_name(scan_mode, kernel_variant):
    """The instantiation rt_stats names, spelled as rocprofv3's kernel trace spells it: render_kernel<MODE, DIAG, SMALLGRID, U53, ITEMS>
    (rt_stats.kernel_variant: bit 0 = the small-grid kernel, bit 1 = 53-bit uniforms, bit 2 = work blocks of 1 024 pixel-samples instead
    of 256; bench.py never sets RT_FLAG_DIAG_STATS)."""
    b = lambda x: "true" if x else "false"
    return f"rt::render_kernel<{int(scan_mode)}, false, {b(kernel_variant & 1)}, {b(kernel_variant & 2)}, {1024 if kernel_variant & 4 else 256}>"


def launch_command(n_ranks, port, argv):
    """The child command of a self-launch: N ranks of this file under torch.distributed.run."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(args):
    """--gpus N > 1 from a bare shell: start the N ranks as a child process and relay its result.
    Nothing in THIS process has touched the GPU (no torch import, no HIP call)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(launch_command(args.gpus, port, sys.argv[1:]), env=env, cwd=ROOT)


def time_config(rt, torch, renderer, flat, w, h, spp, launches, stream, flags=0):
    """Kernel time (HIP events) of `launches` whole-frame launches of one configuration."""
    renderer.upload_scene(flat)
    cam = rt.book1_camera(w, h)
    p = rt.make_params(w, h, spp, seed=1, flags=flags)
    d_fix = torch.zeros((h, w, 3), dtype=torch.int64, device="cuda")
    renderer.render_device(cam, p, d_fix.data_ptr(), stream)      # warm-up
    renderer.last_stats()
    ms = []
    for _ in range(launches):
        renderer.render_device(cam, p, d_fix.data_ptr(), stream)
        st = renderer.last_stats()
        ms.append(st["kernel_ms"])
    del d_fix
    k_ms = float(np.mean(ms))
    return {"n_spheres": int(len(flat)), "width": w, "height": h, "spp": spp, "launches_timed": launches,
            "kernel_ms": round(k_ms, 3), "Msamples_per_s": round(w * h * spp / k_ms / 1e3, 1),
            "rays_per_sample": round(st["rays_traced"] / max(1, st["samples"]), 4),
            "equivalent_bruteforce_TFLOPs": round(st["rays_traced"] * (FLOP_PER_TEST * len(flat) + FLOP_PER_RAY_FIXED)
                                                  / (k_ms * 1e-3) / 1e12, 1)}


def end_to_end(rt, renderer, w, h, spp, calls):
    """SURVEY.md 8(d)'s separate figure: wall time of the HOST-BUFFER calls a maintainer's binding makes for one frame
    (main.rs:122-145: render, to_rgba, flip; scene upload excluded, buffers allocated beforehand, pageable like a Vec<u8>):
    `one_call` = rt_render_rgba8 (sums stay on the device, 4 B/pixel back), `two_calls` = rt_render(out_fix) + rt_resolve_rgba8
    (24 B/pixel out, 24 B/pixel in again, 4 B/pixel out: the path rounds 1-4 documented).  Mean of `calls` frames each, after one
    warm-up frame; kernel_ms = the library's HIP events of the same launches."""
    import ctypes as C
    from rtiow_amd import _ffi
    lib = _ffi.load()
    cam = rt.book1_camera(w, h).to_rt_camera()
    p = rt.make_params(w, h, spp, seed=1)
    rgba = np.zeros((h, w, 4), dtype=np.uint8)
    fix = np.zeros((h, w, 3), dtype=np.uint64)
    st = _ffi.rt_stats()
    vp = lambda a: a.ctypes.data_as(C.c_void_p)

    def one():
        _ffi.check(lib.rt_render_rgba8(renderer._h, C.byref(cam), C.byref(p), 1, vp(rgba), C.byref(st)), "rt_render_rgba8")

    def two():
        _ffi.check(lib.rt_render(renderer._h, C.byref(cam), C.byref(p), None, vp(fix), C.byref(st)), "rt_render")
        _ffi.check(lib.rt_resolve_rgba8(renderer._h, vp(fix), w, h, spp, 1, vp(rgba)), "rt_resolve_rgba8")

    out = {"width": w, "height": h, "spp": spp, "calls_timed": calls}
    crc = {}
    import zlib
    for name, fn in (("one_call", one), ("two_calls", two)):
        fn()
        wall, kern = [], []
        for _ in range(calls):
            t0 = time.perf_counter()
            fn()
            wall.append((time.perf_counter() - t0) * 1e3)
            kern.append(st.kernel_ms)
        crc[name] = zlib.crc32(rgba.tobytes())
        out[name] = {"ms": round(float(np.mean(wall)), 3), "kernel_ms": round(float(np.mean(kern)), 3),
                     "overhead_ms": round(float(np.mean(wall)) - float(np.mean(kern)), 3),
                     "Msamples_per_s": round(w * h * spp / float(np.mean(wall)) / 1e3, 1)}
    out["same_bytes"] = crc["one_call"] == crc["two_calls"]
    return out


def weak_efficiency(ms_per_step, samples_per_gpu, sha, rehearsal):
    """The north_star's second number at N > 1: T(1 GPU) / T(N GPUs) at the same samples per GPU (configs[2] on one
    GPU against configs[4]'s geometry on N).  T(1) is not measured in an N > 1 run: it is read from the newest committed
    `bench.py --weak-baseline` line (profiles/r??_bench_weak_baseline.json) -- a CROSS-RUN ratio, labelled as such; the driver
    forms its own from its back-to-back N = 1, 2, 4, 8 values."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench_weak_baseline.json")))
    if not files:
        return {"value": None, "note": "no profiles/r??_bench_weak_baseline.json to compare with"}
    try:
        base = json.loads(open(files[-1]).read().strip().splitlines()[-1])
        t1 = float(base["ms_per_step"])
        spg = int(base["config"]["samples_per_gpu"])
    except Exception as e:
        return {"value": None, "note": f"unreadable {os.path.basename(files[-1])}: {e}"}
    out = {"value": None, "baseline_ms_per_step": t1, "baseline_samples_per_gpu": spg,
           "baseline_source": os.path.relpath(files[-1], ROOT), "baseline_kernel_source_sha": base["config"].get("kernel_source_sha"),
           "kind": "cross-run: T(N = 1) from the committed --weak-baseline line (another box, another day), T(N) from this run"}
    if rehearsal:
        out["note"] = "rehearsal on one GPU: the ranks share a device, no efficiency is formed"
    elif spg != samples_per_gpu:
        out["note"] = f"this run traces {samples_per_gpu} samples per GPU, the baseline {spg}: not a weak-scaling pair"
    else:
        out["value"] = round(t1 / ms_per_step, 4)
        if out["baseline_kernel_source_sha"] != sha:
            out["note"] = "the baseline line was taken on other kernel sources than this run's"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=0, help="default: 1200 at N=1, 7680 at N>1")
    ap.add_argument("--height", type=int, default=0, help="default: 675 at N=1, 4320 at N>1")
    ap.add_argument("--spp", type=int, default=0,
                    help="samples per pixel PER GPU-share (frame spp = spp x N); default: 500 at N=1, 125 at N>1")
    ap.add_argument("--tile-rows", type=int, default=1,
                    help="rows per shard tile; 1 balances the ranks to within one row of each other")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--weak-baseline", action="store_true",
                    help="N=1 only: the single-GPU half of SURVEY.md 8(e)'s weak-scaling pair -- BASELINE configs[2], "
                         "3840x2160x500 = 4.147 G samples, the per-GPU share of the N>1 runs -- as a whole step "
                         "(render + gather + resolve), so that T(configs[2]) / T(N>1) can be formed from two `value`s")
    ap.add_argument("--tenk", action="store_true",
                    help="N=1 only: BASELINE configs[3] as the timed workload (10k random spheres, 1920x1080, 256 spp) -- "
                         "what tools/profile_bench.sh profiles the large-grid kernel on")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N>1 logic check on a 1-GPU box: every rank renders on cuda:0 and the gather runs "
                         "over gloo on CPU tensors (RCCL refuses two ranks on one device); not a measurement")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    import torch
    import rtiow_amd as rt
    from rtiow_amd.distributed import FrameGatherer, shard_row_map

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    dist = None
    if args.rehearse_on_one_gpu:
        local_rank = 0
    if world > 1:
        # (the rendezvous comes BEFORE anything touches the GPU: a rank that cannot join fails here, on any box)
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # An explicit timeout: a rank that never arrives (or dies inside a step) ends the run non-zero within two
        # minutes instead of holding the others for torch's default (10 min for RCCL, 30 min for gloo).  A step of the
        # N > 1 workload is ~0.6 s; RTIOW_DIST_TIMEOUT_S overrides.
        dist_timeout = datetime.timedelta(seconds=float(os.environ.get("RTIOW_DIST_TIMEOUT_S", "120")))
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo", timeout=dist_timeout)
        else:
            dist.init_process_group("nccl", timeout=dist_timeout, device_id=torch.device(f"cuda:{local_rank}"))
    torch.cuda.set_device(local_rank)

    default_whs = (1200, 675, 500) if world == 1 else (7680, 4320, 125)
    if args.weak_baseline:
        if world != 1:
            sys.exit("--weak-baseline is the N = 1 comparator of the N > 1 runs")
        default_whs = (3840, 2160, 500)
        args.no_other_configs = args.no_cpu_baseline = True
    if args.tenk:
        if world != 1 or args.weak_baseline:
            sys.exit("--tenk is a single-GPU workload")
        default_whs = (1920, 1080, 256)
        args.no_other_configs = True
    W = args.width or default_whs[0]
    H = args.height or default_whs[1]
    spp_share = args.spp or default_whs[2]
    spp_frame = spp_share * world                      # weak scaling: per-GPU samples fixed
    is_default = (W, H, spp_share) == default_whs
    flat = (rt.random_scene(1, grid=(-50, 49)) if args.tenk else rt.random_scene(1)).flatten()
    cam = rt.book1_camera(W, H)
    renderer = rt.Renderer(local_rank)
    renderer.upload_scene(flat)

    params = rt.make_params(W, H, spp_frame, seed=1, max_depth=50, t_min=1e-4,
                            tile_rows=args.tile_rows, shard_index=rank, shard_count=world)
    rows = shard_row_map(H, args.tile_rows, rank, world)
    dev = torch.device(f"cuda:{local_rank}")
    d_fix = torch.zeros((len(rows), W, 3), dtype=torch.int64, device=dev)
    d_rgba = torch.zeros((H, W, 4), dtype=torch.uint8, device=dev) if rank == 0 else None
    stream = torch.cuda.current_stream().cuda_stream
    rehearse = args.rehearse_on_one_gpu and world > 1
    gather = FrameGatherer(H, W, args.tile_rows, rank, world, "cpu" if rehearse else dev)

    kernel_ms, gather_ms, rays, samples = [], [], 0, 0
    gather_events = []

    def step(record):
        nonlocal rays, samples
        renderer.render_device(cam, params, d_fix.data_ptr(), stream)
        if rehearse:
            tg = time.perf_counter()
            full = gather(d_fix.cpu())
            full = full.to(dev) if rank == 0 else None
            if record:
                gather_ms.append((time.perf_counter() - tg) * 1e3)
        else:
            # (events on the launch stream: the collective runs on RCCL's stream, which the launch stream waits for)
            ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ea.record()
            full = gather(d_fix)
            eb.record()
            if record:
                gather_events.append((ea, eb))
        if rank == 0:
            renderer.resolve_rgba8_device(full.data_ptr(), W, H, spp_frame, 1, d_rgba.data_ptr(), stream)
            step.last_full = full
        if record:
            st = renderer.last_stats()                 # waits for this launch's events only
            kernel_ms.append(st["kernel_ms"])
            rays, samples = st["rays_traced"], st["samples"]
            step.scan_mode = st["scan_mode"]
            step.kernel_variant = st.get("kernel_variant", 0)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    step.last_full = None
    step.scan_mode = -1
    step.kernel_variant = 0

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t0
    gather_ms += [ea.elapsed_time(eb) for ea, eb in gather_events]
    per_rank = None
    if dist is not None:
        # max over ranks of the timed region; every rank's own kernel and gather time (mean per step) for the record
        t = torch.tensor([elapsed, float(np.mean(kernel_ms)), float(np.mean(gather_ms))], dtype=torch.float64,
                         device="cpu" if args.rehearse_on_one_gpu else dev)
        parts = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        elapsed = max(float(q[0].item()) for q in parts)
        per_rank = {"elapsed_s": [round(float(q[0].item()), 6) for q in parts],
                    "kernel_ms": [round(float(q[1].item()), 3) for q in parts],
                    "gather_ms": [round(float(q[2].item()), 3) for q in parts]}

    if rank == 0:
        frame_samples = W * H * spp_frame
        value = frame_samples * args.steps / elapsed / 1e6
        k_ms = float(np.mean(kernel_ms))
        n_sph = int(len(flat))
        flops = rays * (FLOP_PER_TEST * n_sph + FLOP_PER_RAY_FIXED)        # this rank's launch
        achieved = flops / (k_ms * 1e-3) / 1e12
        algo_bytes = len(rows) * W * 12 + n_sph * 36
        # Counters that need rocprofv3 --pmc cannot be measured inside this process: they are REPLAYED
        # from the committed passes of the same command (profiles/pmc_replay.json, written by
        # tools/summarize_profile.py) and only when the kernel sources are the profiled ones.
        sha = kernel_source_sha()
        from rtiow_amd import _ffi
        lib_sha = _ffi.load().rt_build_source_sha().decode()
        pmc, pmc_note = None, "no profiles/pmc_replay.json entry for this configuration"
        path = os.path.join(ROOT, "profiles", "pmc_replay.json")
        if os.path.exists(path):
            try:
                for entry in json.load(open(path)).get("entries", []):
                    if entry.get("config") == [W, H, spp_frame, world]:
                        if lib_sha != sha:
                            pmc_note = (f"the loaded library was built from kernel sources {lib_sha}, the tree is {sha}: "
                                        f"rebuild (./build_lib.sh); no counters replayed")
                        elif entry.get("kernel_source_sha") == sha:
                            pmc, pmc_note = entry, entry.get("source")
                        else:
                            pmc_note = (f"stale: {entry.get('source')} was taken on kernel sources "
                                        f"{entry.get('kernel_source_sha')}, this run is {sha}")
            except Exception as e:                      # a broken replay file must not break the bench
                pmc_note = f"unreadable profiles/pmc_replay.json: {e}"
        frame_crc = None
        if step.last_full is not None:
            import zlib
            frame_crc = zlib.crc32(step.last_full.cpu().numpy().tobytes()) & 0xFFFFFFFF
        roof = {
            # What bounds the kernel is vector-instruction ISSUE.
            #   achieved = vector (VALU) wave-instructions issued per second: the instruction count of one launch
            #              (SQ_INSTS_VALU, replayed: a property of kernel + configuration) / this run's kernel time;
            #   peak     = what the hardware can issue for THIS instruction mix: 1024 SIMDs x 2.4 GHz / (mean hardware
            #              issue cycles per instruction: 2 for f32/int/bit, 4 for f64 add/mul/fma, 8 for the
            #              transcendental unit -- the class counts are replayed PMC counters, the cycles are
            #              MI355X_MICROARCH.md's constants); `peak_uniform` = 1024 x 2.4 / 2, every instruction at 2 cycles;
            #   frac     = achieved / peak;  `valu_busy` = the measured share of SIMD cycles in which a vector
            #              instruction is issuing (SQ_ACTIVE_INST_VALU x 4 / (SIMDs x cycles)) -- a utilisation, not a roofline.
            # (all four template arguments <MODE, DIAG, SMALLGRID, U53>: the name rocprofv3's kernel trace prints)
            "bound": "valu-issue", "kernel": kernel_name(step.scan_mode, step.kernel_variant),
            "achieved": None, "peak": None, "unit": "G vector wave-instructions/s", "frac": None,
            "kernel_ms": round(k_ms, 3), "kernel_ms_source": "HIP events on the launch stream, this run",
            "launches_timed": len(kernel_ms),
            "traffic": None, "issue": None, "mfma": None, "counters_source": pmc_note,
            # SURVEY.md 8(d)'s yardstick, kept for comparison with round 1: the flop a BRUTE-FORCE f32 scan would do for
            # the rays traced (17 per ray-sphere pair + 65 per ray) over this kernel's time.  It is not flop the kernel
            # executes -- the scan is a bf16 matrix-pipe filter over the tiles a wave's rays can reach -- so it may
            # exceed the FP32 vector peak: a speed-up over brute force, not a roofline.
            "bruteforce_equivalent": {"TFLOPs": round(achieved, 3), "flop_per_launch": flops,
                                      "fp32_vector_peak_TFLOPs": PEAK_FP32_VECTOR_TFLOPS,
                                      "ratio_to_fp32_vector_peak": round(achieved / PEAK_FP32_VECTOR_TFLOPS, 4)},
            "hbm": {"algorithmic_bytes_per_launch": algo_bytes,
                    "achieved_GBps": round(algo_bytes / (k_ms * 1e-3) / 1e9, 4),
                    "peak_GBps": PEAK_HBM_GBPS,
                    "frac": round(algo_bytes / (k_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, 8)},
        }
        if pmc is not None:
            roof["traffic"] = pmc.get("hbm_bytes_per_launch")
            if pmc.get("rocprof_avg_kernel_ms"):
                roof["kernel_ms_rocprof_avg"] = pmc["rocprof_avg_kernel_ms"]
            if pmc.get("valu_insts_per_launch") and pmc.get("simd_cycles_per_launch"):
                wave_bounces = rays / 64.0
                n_valu = pmc["valu_insts_per_launch"]
                rate = n_valu / (k_ms * 1e-3) / 1e9
                n_f64 = pmc.get("f64_addmulfma_insts_per_launch") or 0.0
                n_trans = pmc.get("trans_f32_insts_per_launch") or 0.0
                mean_cycles = ((n_valu - n_f64 - n_trans) * ISSUE_CYCLES["plain"] + n_f64 * ISSUE_CYCLES["f64"]
                               + n_trans * ISSUE_CYCLES["trans"]) / n_valu
                peak = N_SIMD * SPEC_CLOCK_GHZ / mean_cycles
                roof["achieved"] = round(rate, 1)
                roof["peak"] = round(peak, 1)
                roof["frac"] = round(rate / peak, 4)
                roof["peak_uniform"] = round(N_SIMD * SPEC_CLOCK_GHZ / ISSUE_CYCLES["plain"], 1)
                roof["frac_of_peak_uniform"] = round(rate / roof["peak_uniform"], 4)
                roof["peak_source"] = (f"{N_SIMD} SIMDs x {SPEC_CLOCK_GHZ} GHz / {mean_cycles:.3f} hardware issue cycles per "
                                       f"instruction of this mix ({n_f64 / n_valu:.3f} f64 add/mul/fma at 4, "
                                       f"{n_trans / n_valu:.4f} transcendental at 8, the rest at 2)")
                roof["valu_busy"] = pmc.get("valu_busy")
                # `frac` prices whatever the kernel issues: an optimisation that REMOVES instructions lowers it (round 5: 0.53 -> 0.51 while the
                # frame got 2 % faster).  The figure that cannot be gamed that way: the share of the kernel's time that the f64 add/mul/fma the
                # reference's arithmetic REQUIRES (its divisions' and square roots' expansions included) would take alone at the hardware's f64
                # issue rate (4 cycles per wave-instruction) -- everything else the kernel issues is this implementation's overhead.
                if n_f64:
                    roof["frac_required_f64_only"] = round(n_f64 * ISSUE_CYCLES["f64"] / (N_SIMD * SPEC_CLOCK_GHZ * 1e9) / (k_ms * 1e-3), 4)
                roof["issue"] = {
                    "valu_insts_per_launch": n_valu,
                    "valu_insts_per_wave_bounce": round(n_valu / wave_bounces, 1),
                    # the f64 add/mul/fma the reference's arithmetic REQUIRES (SQ_INSTS_VALU_{ADD,MUL,FMA}_F64): the rest of the
                    # vector instructions are overhead of this implementation (Philox, the look, lists, conversions, moves)
                    "required_f64_per_wave_bounce": round(n_f64 / wave_bounces, 1) if n_f64 else None,
                    "salu_insts_per_wave_bounce": (None if not pmc.get("salu_insts_per_launch") else
                                                   round(pmc["salu_insts_per_launch"] / wave_bounces, 1)),
                    "branches_per_wave_bounce": (None if not pmc.get("branch_insts_per_launch") else
                                                 round(pmc["branch_insts_per_launch"] / wave_bounces, 1)),
                    "mfma_insts_per_wave_bounce": (None if not pmc.get("mfma_insts_per_launch") else
                                                   round(pmc["mfma_insts_per_launch"] / wave_bounces, 1)),
                    "lds_bank_conflict_share": pmc.get("lds_bank_conflict_share"),
                    # SQ_ACTIVE_INST_VALU (quad-cycles) x 4 / (SIMDs x kernel cycles): share of SIMD time in which
                    # a vector instruction is being issued
                    "valu_busy": pmc.get("valu_busy"),
                    "cycles_per_valu_inst": round(pmc["simd_cycles_per_launch"] / (n_valu / N_SIMD), 3),
                }
            if pmc.get("mfma_insts_per_launch"):
                t = pmc["mfma_insts_per_launch"] * 32768 / (k_ms * 1e-3) / 1e12     # v_mfma_f32_32x32x16_bf16 = 32768 flop
                roof["mfma"] = {"achieved_TFLOPs": round(t, 1), "peak_TFLOPs": PEAK_BF16_MFMA_TFLOPS,
                                "frac": round(t / PEAK_BF16_MFMA_TFLOPS, 4)}
        scene = (f"10k random spheres (random_scene seed 1 over a, b in -50..49, {n_sph} spheres)" if args.tenk else
                 f"book-1 final scene (random_scene seed 1, {n_sph} spheres)")
        if not is_default:
            tag = "[custom size]"
        elif args.tenk:
            tag = "[BASELINE.json configs[3]]"
        elif args.weak_baseline:
            tag = "[BASELINE.json configs[2]: the N = 1 half of the weak-scaling pair, 4.147 G samples on one GPU]"
        elif world == 1:
            tag = "[the north_star target configuration: BASELINE.json configs[1]'s frame at 500 spp]"
        else:
            tag = (f"[BASELINE.json configs[4]'s geometry, 125 spp per GPU: {world}/8 of configs[4]; "
                   f"4.147 G samples per GPU = configs[2] on one GPU]")
        workload = f"{scene}, {W}x{H}, {spp_frame} spp, depth 50 {tag}"
        out = {
            "metric": baseline_metric(),
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": workload,
                "width": W, "height": H, "spp": spp_frame, "max_depth": 50, "n_spheres": n_sph,
                "samples_per_gpu": frame_samples // world,
                "sharding": "whole frame on one GPU" if world == 1 else
                            f"row tiles of {args.tile_rows} dealt round-robin to {world} ranks, one RCCL gather"
                            + (" (REHEARSAL: all ranks on cuda:0, gloo gather)" if rehearse else ""),
                "rays_per_sample": round(rays / max(1, samples), 4),
                "frame_crc32": frame_crc,      # of the exact sums: equal for equal (W, H, spp) at any N
                "kernel_source_sha": sha, "library_source_sha": lib_sha,
            },
            "roofline": roof,
        }
        if world == 1 and not args.no_other_configs:
            # the other single-GPU BASELINE configurations, a few launches each (kernel time, HIP events)
            others = []
            try:
                others.append({"config": "configs[0] book-1 400x225x10 (the reference's own CPU-runnable case)", **time_config(rt, torch, renderer, flat, 400, 225, 10, 10, stream)})
                others.append({"config": "configs[1] book-1 1200x675x100", **time_config(rt, torch, renderer, flat, 1200, 675, 100, 10, stream)})
                # the price of the reference's exact bit count: every draw from TWO Philox words (53 random bits, rand 0.8.5's gen::<f64>())
                # instead of one word's 32 -- another valid stream, the same frame statistically
                others.append({"config": "configs[1] book-1 1200x675x100 with RT_FLAG_UNIFORM53 (53-bit draws)",
                               **time_config(rt, torch, renderer, flat, 1200, 675, 100, 10, stream, flags=rt.RT_FLAG_UNIFORM53)})
                others.append({"config": "configs[2] book-1 3840x2160x500", **time_config(rt, torch, renderer, flat, 3840, 2160, 500, 2, stream)})
                tenk = rt.random_scene(1, grid=(-50, 49)).flatten()
                c4 = time_config(rt, torch, renderer, tenk, 1920, 1080, 256, 2, stream)
                # (no fraction of the FP32 vector peak here: the scan runs on the bf16 matrix pipe, and on this
                #  scene the brute-force-equivalent rate exceeds that peak -- it is a speed-up, not a roofline)
                others.append({"config": "configs[3] 10k spheres 1920x1080x256", **c4})
            except Exception as e:
                others.append({"error": str(e)})
            renderer.upload_scene(flat)
            out["other_configs"] = others
        if world == 1 and is_default and not args.weak_baseline and not args.tenk and not args.no_end_to_end:
            # the end-to-end figure through the host-buffer entry points (PCIe-inclusive; never `value`): what the
            # reference's main() would see for the headline frame and for BASELINE configs[1]
            try:
                renderer.upload_scene(flat)
                out["end_to_end"] = {
                    "what": "wall ms per frame of the host-buffer C-ABI calls replacing main.rs:122-145 (scene upload excluded; "
                            "pageable host buffers): one_call = rt_render_rgba8, two_calls = rt_render + rt_resolve_rgba8",
                    "headline": end_to_end(rt, renderer, W, H, spp_frame, 5),
                    "configs[1]": end_to_end(rt, renderer, 1200, 675, 100, 10),
                    "configs[2]": end_to_end(rt, renderer, 3840, 2160, 500, 2),      # (33 MB of RGBA8; the sums are 199 MB)
                }
            except Exception as e:
                out["end_to_end"] = {"error": str(e)}
        if per_rank is not None:
            out["per_rank"] = per_rank
        if world > 1:
            out["weak_efficiency"] = weak_efficiency(elapsed / args.steps * 1e3, frame_samples // world, sha, rehearse)
        out["gather_ms"] = round(float(np.mean(gather_ms)), 3)       # rank 0, mean per timed step (at N = 1: the copy into frame order)
        out["rmse_vs_cpu"] = None
        if world == 1 and not args.no_cpu_baseline:
            # (the 10k-sphere scene is ~20x slower per sample on the CPU: fewer rows there)
            cb, rows_j, cpu_sums = cpu_baseline(flat, W, H, spp_frame, min_rows=6 if args.tenk else 100)
            out["cpu_baseline"] = cb
            out["cpu_baseline"]["gpu_over_cpu"] = round(value / cb["value"], 1)
            # the RMSE half of the metric, in the same run: the rows Oracle A has just rendered against the same rows of
            # the frame the last timed step left on the GPU (exact sums and resolved RGBA8 bytes)
            idx = torch.as_tensor(rows_j, device=dev)
            gpu_rows = step.last_full.index_select(0, idx).cpu().numpy().view(np.uint64)
            gpu_rgba = d_rgba.index_select(0, (H - 1) - idx).cpu().numpy()          # flipped frame: image row y = H-1-j
            par = rmse_vs_cpu(gpu_rows, gpu_rgba, cpu_sums, spp_frame)
            out["rmse_vs_cpu"] = par["rmse"]
            out["rmse_gate"] = 1e-4
            out["parity_vs_cpu"] = {
                **par, "rows": int(len(rows_j)), "row_stride": int(rows_j[1] - rows_j[0]) if len(rows_j) > 1 else 0,
                "pixels": int(len(rows_j)) * W, "samples": int(len(rows_j)) * W * spp_frame,
                "definition": "sqrt(mean over pixels and channels of (GPU_mean - CPU_mean)^2) on linear radiance means "
                              "(main.rs:135-137: pixel_color / spp, before gamma); CPU = Oracle A (literal f64 "
                              "restatement), same seed, same rows of the frame timed above; rgba8 = Color::to_rgba bytes",
            }
        print(json.dumps(out), flush=True)

    renderer.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
