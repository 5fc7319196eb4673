#!/usr/bin/env python3
"""bench.py -- Msamples/s (pixels x spp) of the render path on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one complete frame of BASELINE.json configs[1] (book-1 final scene,
1200x675, depth 50): every rank renders its interleaved row tiles with inputs
resident in HBM, ONE gather (RCCL) brings the exact sums to rank 0, rank 0
resolves them to RGBA8 (Color::to_rgba + flip).  Weak scaling: the frame keeps
its geometry and spp = 100 x N, so every GPU traces the same number of
pixel-samples at any N.  Prints ONE JSON line on rank 0.

The JSON also carries
  roofline      for the dominant kernel (render_kernel): algorithmic FLOPs per
                launch / its mean duration (HIP events recorded by the library
                on the launch stream), against the FP32 vector peak -- the scan
                is VALU-bound, neither HBM- nor MFMA-bound;
  cpu_baseline  Oracle A (the literal f64 CPU restatement of the reference,
                oracle/oracle_f64.c) timed on this host's cores on a bounded row
                subset of the same frame (rank 0, N=1 only).
"""
import argparse
import json
import os
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


def cpu_baseline(flat, width, height, spp, target_seconds=12.0):
    """Oracle A on a bounded, evenly strided row subset of the same frame."""
    import oracle
    cam = oracle.book1_camera(width, height)
    threads = host_cpu_share()
    probe_step = 64
    p = oracle.make_params(width, height, spp, rows=(0, height, probe_step), nthreads=threads)
    _, st = oracle.render_a(cam, flat, p)
    rate = st["samples"] / st["seconds"]
    rows_wanted = max(1.0, target_seconds * rate / (width * spp))
    step = int(min(probe_step, max(1, round(height / rows_wanted))))
    p = oracle.make_params(width, height, spp, rows=(0, height, step), nthreads=threads)
    _, st = oracle.render_a(cam, flat, p)
    nrows = oracle.n_rows(p)
    return {
        "value": round(st["samples"] / st["seconds"] / 1e6, 4), "unit": "Msamples/s",
        "cores": int(st["threads_used"]), "kind": "port",
        "sample": f"Oracle A (literal f64 restatement, oracle/oracle_f64.c), rows j=0,{step},2*{step},... "
                  f"({nrows} of {height} rows) of the {width}x{height}x{spp}spp frame, {st['samples']} samples in "
                  f"{st['seconds']:.2f} s on {st['threads_used']} host threads",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=1200)
    ap.add_argument("--height", type=int, default=675)
    ap.add_argument("--spp", type=int, default=100, help="samples per pixel PER GPU-share (frame spp = spp x N)")
    ap.add_argument("--tile-rows", type=int, default=1,
                    help="rows per shard tile; 1 balances the ranks to within one row of each other")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N>1 logic check on a 1-GPU box: every rank renders on cuda:0 and the gather runs "
                         "over gloo on CPU tensors (RCCL refuses two ranks on one device); not a measurement")
    args = ap.parse_args()

    import torch
    import rtiow_amd as rt
    from rtiow_amd.distributed import FrameGatherer, shard_row_map

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if args.gpus != 1 and world == 1:
            raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run --nproc-per-node {args.gpus}")
        args.gpus = world
    dist = None
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))

    W, H = args.width, args.height
    spp_frame = args.spp * world                       # weak scaling: per-GPU samples fixed
    flat = rt.random_scene(1).flatten()
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

    kernel_ms, rays, samples = [], 0, 0

    def step(record):
        nonlocal rays, samples
        renderer.render_device(cam, params, d_fix.data_ptr(), stream)
        if rehearse:
            full = gather(d_fix.cpu())
            full = full.to(dev) if rank == 0 else None
        else:
            full = gather(d_fix)
        if rank == 0:
            renderer.resolve_rgba8_device(full.data_ptr(), W, H, spp_frame, 1, d_rgba.data_ptr(), stream)
            step.last_full = full
        if record:
            st = renderer.last_stats()                 # waits for this launch's events only
            kernel_ms.append(st["kernel_ms"])
            rays, samples = st["rays_traced"], st["samples"]
            step.scan_mode = st["scan_mode"]

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    step.last_full = None
    step.scan_mode = -1

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.rehearse_on_one_gpu else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        frame_samples = W * H * spp_frame
        value = frame_samples * args.steps / elapsed / 1e6
        k_ms = float(np.mean(kernel_ms))
        n_sph = int(len(flat))
        flops = rays * (FLOP_PER_TEST * n_sph + FLOP_PER_RAY_FIXED)        # this rank's launch
        achieved = flops / (k_ms * 1e-3) / 1e12
        algo_bytes = len(rows) * W * 12 + n_sph * 36
        # counters of the same launch configuration from the committed rocprofv3 --pmc passes (profiles/)
        traffic = mfma_insts = valu_busy = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                j = json.load(open(pmc))
                if j.get("config") == [W, H, spp_frame, world]:
                    traffic = j.get("hbm_bytes_per_launch")
                    mfma_insts = j.get("mfma_insts_per_launch")
                    valu_busy = j.get("valu_busy")
            except Exception:
                traffic = mfma_insts = valu_busy = None
        frame_crc = None
        if step.last_full is not None:
            import zlib
            frame_crc = zlib.crc32(step.last_full.cpu().numpy().tobytes()) & 0xFFFFFFFF
        out = {
            "metric": "Msamples/sec (pixels x spp) on book-1 final scene",
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": f"book-1 final scene (random_scene seed 1, {n_sph} spheres), {W}x{H}, "
                            f"{spp_frame} spp, depth 50 [BASELINE.json configs[1]"
                            + ("" if world == 1 else f", spp scaled x{world} for weak scaling") + "]",
                "width": W, "height": H, "spp": spp_frame, "max_depth": 50, "n_spheres": n_sph,
                "samples_per_gpu": frame_samples // world,
                "sharding": "whole frame on one GPU" if world == 1 else
                            f"row tiles of {args.tile_rows} dealt round-robin to {world} ranks, one RCCL gather",
                "rays_per_sample": round(rays / max(1, samples), 4),
                "frame_crc32": frame_crc,      # of the exact sums: equal for equal (W, H, spp) at any N
            },
            "roofline": {
                "bound": "valu", "kernel": f"rt::render_kernel<{step.scan_mode}, false>",
                "achieved": round(achieved, 3), "peak": PEAK_FP32_VECTOR_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_FP32_VECTOR_TFLOPS, 4),
                "traffic": traffic,
                "kernel_ms": round(k_ms, 3), "launches_timed": len(kernel_ms),
                "algorithmic_flop_per_launch": flops,
                # the matrix pipe's own view: v_mfma_f32_32x32x16_bf16 = 32768 flop each, against the dense bf16 peak
                "mfma": (None if mfma_insts is None else
                         {"achieved_TFLOPs": round(mfma_insts * 32768 / (k_ms * 1e-3) / 1e12, 1), "peak_TFLOPs": PEAK_BF16_MFMA_TFLOPS,
                          "frac": round(mfma_insts * 32768 / (k_ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4)}),
                "valu_busy": None if valu_busy is None else round(valu_busy, 3),
                "hbm": {"algorithmic_bytes_per_launch": algo_bytes,
                        "achieved_GBps": round(algo_bytes / (k_ms * 1e-3) / 1e9, 4),
                        "peak_GBps": PEAK_HBM_GBPS,
                        "frac": round(algo_bytes / (k_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, 8)},
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(flat, W, H, spp_frame)
            out["cpu_baseline"]["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)

    renderer.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
