"""Diagnostic: cfg2 kernel time against workgroups per CU (RTIOW_BLOCKS_PER_CU): how latency-bound is the bounce loop?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import rtiow_amd as rt
flat = rt.random_scene(1).flatten()
for bpc in (1, 2, 3, 4):
    os.environ["RTIOW_BLOCKS_PER_CU"] = str(bpc)
    r = rt.Renderer(0); r.upload_scene(flat)
    best = 1e9
    for _ in range(3):
        sm, fix, st = r.render(rt.book1_camera(1200, 675), rt.make_params(1200, 675, 100), want_fix=False)
        best = min(best, st["kernel_ms"])
    print(f"workgroups/CU {bpc} (waves/SIMD {bpc}): {best:.2f} ms  grid {st['grid_blocks']}", flush=True)
    r.close()
