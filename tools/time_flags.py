"""Diagnostic: cfg2 kernel time with and without RT_FLAG_UNIFORM53 (median of interleaved launches)."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import rtiow_amd as rt
r = rt.Renderer(0)
r.upload_scene(rt.random_scene(1).flatten())
w, h, spp = 1200, 675, 100
cam = rt.book1_camera(w, h)
ts = {0: [], rt.RT_FLAG_UNIFORM53: []}
for rnd in range(6):
    for f in ts:
        _, _, st = r.render(cam, rt.make_params(w, h, spp, flags=f), want_fix=False)
        if rnd:
            ts[f].append(st["kernel_ms"])
for f, v in ts.items():
    print(f"flags {f:#x}: median {statistics.median(v):.3f} ms ({w}x{h}x{spp}, rays/sample {st['rays_traced'] / st['samples']:.3f})")
r.close()
