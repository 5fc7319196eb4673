"""Diagnostic: median kernel time per scan mode (MODES env) on cfg2, and one launch of cfg4 (10k spheres) if CFG4=1."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import rtiow_amd as rt
modes = [int(x) for x in os.environ.get("MODES", "5,4,3").split(",")]
flat = rt.random_scene(1).flatten()
for rep in range(2):
    for mode in modes:
        os.environ["RTIOW_SCAN_MODE"] = str(mode)
        r = rt.Renderer(0); r.upload_scene(flat)
        ts = []
        for _ in range(7):
            sm, fix, st = r.render(rt.book1_camera(1200, 675), rt.make_params(1200, 675, 100), want_fix=False)
            ts.append(st["kernel_ms"])
        print(f"cfg2 mode {mode}: median {statistics.median(ts[1:]):.3f} ms  min {min(ts):.3f}", flush=True)
        r.close()
if os.environ.get("CFG4"):
    big = rt.random_scene(1, grid=(-50, 49)).flatten()
    for mode in modes:
        os.environ["RTIOW_SCAN_MODE"] = str(mode)
        r = rt.Renderer(0); r.upload_scene(big)
        sm, fix, st = r.render(rt.book1_camera(1920, 1080), rt.make_params(1920, 1080, 16, flags=rt.RT_FLAG_DIAG_STATS), want_fix=False)
        print(f"cfg4 (16 spp) mode {mode}: {st['kernel_ms']:.1f} ms  {st['samples'] / st['kernel_ms'] / 1e3:.1f} Msamples/s  cand/ray {st['candidates'] / st['rays_traced']:.2f} roots/ray {st['exact_roots'] / st['rays_traced']:.2f}", flush=True)
        r.close()
