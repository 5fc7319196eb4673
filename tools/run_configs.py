"""Runs BASELINE.json's single-GPU configurations once each (timing + sanity), prints JSON lines."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtiow_amd as rt

cfgs = [
    ("configs[0] book-1 400x225x10", 1, (-11, 11), 400, 225, 10),
    ("configs[1] book-1 1200x675x100", 1, (-11, 11), 1200, 675, 100),
    ("north_star target book-1 1200x675x500", 1, (-11, 11), 1200, 675, 500),
    ("configs[2] book-1 3840x2160x500", 1, (-11, 11), 3840, 2160, 500),
    ("configs[3] 10k spheres 1920x1080x256", 1, (-50, 49), 1920, 1080, 256),
]
r = rt.Renderer(0)
for name, seed, grid, w, h, spp in cfgs:
    flat = rt.random_scene(seed, grid=grid).flatten()
    r.upload_scene(flat)
    cam = rt.book1_camera(w, h)
    t0 = time.perf_counter()
    sm, fix, st = r.render(cam, rt.make_params(w, h, spp), want_fix=False)            # the timed, shipped path
    wall = time.perf_counter() - t0
    _, _, sd = r.render(cam, rt.make_params(w, h, max(1, spp // 20), flags=rt.RT_FLAG_DIAG_STATS), want_fix=False)
    mean = sm.astype(np.float64).mean() / spp
    n = len(flat)
    flops = st["rays_traced"] * (17 * n + 65)
    print(json.dumps({"config": name, "n_spheres": n, "samples": st["samples"], "kernel_ms": round(st["kernel_ms"], 2),
                      "wall_ms_incl_d2h": round(wall * 1e3, 1), "Msamples_per_s": round(w * h * spp / st["kernel_ms"] / 1e3, 1),
                      "rays_per_sample": round(st["rays_traced"] / st["samples"], 3),
                      "cand_per_ray": round(sd["candidates"] / sd["rays_traced"], 2),
                      "algorithmic_TFLOPs": round(flops / st["kernel_ms"] / 1e9, 1), "mean_radiance": round(mean, 4)}), flush=True)
r.close()
