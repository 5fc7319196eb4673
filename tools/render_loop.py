"""Diagnostic workload: N whole-frame launches of one configuration through rt_render_device (nothing else on the GPU), for
profilers that want a plain stream of render kernels (PC sampling, PMC passes).
usage: python3 tools/render_loop.py [W H SPP N] [--tenk] [--u53]      default 1200 675 100 10"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtiow_amd as rt
a = [x for x in sys.argv[1:] if not x.startswith("--")]
W, H, SPP, N = (int(a[0]), int(a[1]), int(a[2]), int(a[3])) if len(a) >= 4 else (1200, 675, 100, 10)
flat = (rt.random_scene(1, grid=(-50, 49)) if "--tenk" in sys.argv else rt.random_scene(1)).flatten()
r = rt.Renderer(0)
r.upload_scene(flat)
cam = rt.book1_camera(W, H)
p = rt.make_params(W, H, SPP, seed=1, flags=rt.RT_FLAG_UNIFORM53 if "--u53" in sys.argv else 0)
d_fix = torch.zeros((H, W, 3), dtype=torch.int64, device="cuda")
ms = []
for _ in range(N):
    r.render_device(cam, p, d_fix.data_ptr(), torch.cuda.current_stream().cuda_stream)
    st = r.last_stats()
    ms.append(st["kernel_ms"])
print(f"{W}x{H}x{SPP} x{N}: kernel ms min {min(ms):.3f} median {sorted(ms)[len(ms)//2]:.3f}; rays {st['rays_traced']} variant {st['kernel_variant']}")
