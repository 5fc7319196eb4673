import time, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import rtiow_amd as rt
r = rt.Renderer(0)
for name, grid in (("book", (-11, 11)), ("10k", (-50, 49)), ("40k", (-100, 99)), ("90k", (-150, 149))):
    flat = rt.random_scene(1, grid=grid).flatten()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); r.upload_scene(flat); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{name}: {len(flat)} spheres, rt_upload_scene {min(ts):.2f} ms (best of 3; {ts})")
    w, h = 320, 180
    t0 = time.perf_counter(); _, _, st = r.render(rt.book1_camera(w, h), rt.make_params(w, h, 16)); t1 = time.perf_counter()
    print(f"   render {w}x{h}x16: kernel {st['kernel_ms']:.2f} ms, call {1e3 * (t1 - t0):.2f} ms, variant {st['kernel_variant']}")
