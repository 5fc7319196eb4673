"""Diagnostic: kernel time of configs[1] and configs[3] (at 64 spp) for several grid resolutions (RTIOW_GRID_DIM = cells
per side of the tile grid, 0 = the library's own choice).  usage: python tools/grid_dim_sweep.py 0 4 5 18 19 20"""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, statistics
    import rtiow_amd as rt
    out = {}
    r = rt.Renderer(0)
    for name, grid, w, h, spp in [("cfg2", (-11, 11), 1200, 675, 100), ("cfg4", (-50, 49), 1920, 1080, 64)]:
        r.upload_scene(rt.random_scene(1, grid=grid).flatten())
        ts = []
        for _ in range(4):
            _, _, st = r.render(rt.book1_camera(w, h), rt.make_params(w, h, spp), want_fix=False)
            ts.append(st["kernel_ms"])
        out[name] = round(statistics.median(ts[1:]), 3)
    print(json.dumps(out))
else:
    for f in sys.argv[1:]:
        env = dict(os.environ, RTIOW_GRID_DIM=f)
        p = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        print("grid_dim", f, p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-300:], flush=True)
