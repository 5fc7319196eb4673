"""Host-buffer entry point (rt_render): wall time including the D2H copy of the frame."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtiow_amd as rt
r = rt.Renderer(0); r.upload_scene(rt.random_scene(1).flatten())
cam = rt.book1_camera(1200, 675); p = rt.make_params(1200, 675, 100)
for want_fix in (False, True):
    r.render(cam, p, want_fix=want_fix)
    t0 = time.perf_counter(); n = 5
    for _ in range(n): sm, fix, st = r.render(cam, p, want_fix=want_fix)
    dt = (time.perf_counter() - t0) / n
    print(f"rt_render host buffers (f32 sums{' + u64 sums' if want_fix else ''}): {dt*1e3:.2f} ms wall per frame = {81e6/dt/1e6:.0f} Msamples/s (kernel alone {st['kernel_ms']:.2f} ms)")
