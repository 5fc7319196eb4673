"""Randomised GPU-vs-oracle parity soak (development tool): many random scenes, cameras, image
sizes and seeds; every frame must equal Oracle B bit for bit.  Usage: fuzz_parity.py [cases] [seed0]
FUZZ_LARGE=p: share of scenes with enough spheres for a grid of more than 64 cells (default 0.15);
FUZZ_U53=p: share of cases rendered with RT_FLAG_UNIFORM53 (default 0); FUZZ_HIGH_SPP=p: share of cases with 69..400 samples per
pixel on a tiny image (with RTIOW_LARGE_BLOCK_MIN_ITEMS=0 those launches use the work blocks of 1 024 pixel-samples)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
import rtiow_amd as rt

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
r = rt.Renderer(0)
bad = 0
t0 = time.time()
tot_rays = 0
for case in range(cases):
    rng = np.random.default_rng(seed0 * 100003 + case)
    n = int(rng.integers(1, 1200))
    if rng.random() < float(os.environ.get("FUZZ_LARGE", "0.15")):         # enough spheres for a grid of more than 64 cells (the kernel's second way of listing tiles)
        n = int(rng.integers(1200, 5000))
    spread = float(10.0 ** rng.uniform(0.0, 3.0))
    w = rt.HittableList()
    if rng.random() < 0.6:
        w.push(rt.Sphere(rt.Point3(0, -1000 * spread / 10, 0), 1000 * spread / 10, rt.Lambertian(rt.Color(0.5, 0.5, 0.5))))
    rscale = spread / 10 * float(10.0 ** rng.uniform(-1.5, 0.3))
    clusters = rng.uniform(-spread, spread, (int(rng.integers(1, 6)), 3)) if rng.random() < 0.25 else None
    for _ in range(n):
        c = rng.uniform(-spread, spread, 3)
        if clusters is not None and rng.random() < 0.85:     # dense clumps: grid cells overflow into the tiles every ray scans
            c = clusters[rng.integers(0, len(clusters))] + rng.normal(size=3) * spread * 0.05
        c[1] = abs(c[1]) * rng.uniform(0.0, 0.5)
        rad = float(rng.uniform(0.2, 1.5)) * rscale
        k = rng.integers(0, 3)
        m = (rt.Lambertian(rng.uniform(0.05, 0.95, 3)) if k == 0 else
             rt.Metal(rng.uniform(0.5, 1.0, 3), float(rng.uniform(0.0, 0.5))) if k == 1 else
             rt.Dialectric(float(rng.uniform(1.1, 2.5))))
        w.push(rt.Sphere(c, rad, m))
    flat = w.flatten()
    W, H, spp = int(rng.integers(8, 64)), int(rng.integers(6, 48)), int(rng.integers(1, 5))
    if rng.random() < 0.4:          # launches of >= 5 / 9 spp keep their work blocks' sums in LDS (a different write path; blocks of 64 .. 256 pixel-samples)
        W, H, spp = int(rng.integers(6, 28)), int(rng.integers(5, 20)), int(rng.integers(5, 90))
    if rng.random() < float(os.environ.get("FUZZ_HIGH_SPP", "0")):
        W, H, spp = int(rng.integers(5, 20)), int(rng.integers(4, 14)), int(rng.integers(69, 400))
    lf = rng.uniform(-spread, spread, 3); lf[1] = abs(lf[1]) * 0.3 + 0.3 * spread / 10
    la = rng.uniform(-spread, spread, 3) * 0.3
    cam = rt.Camera(lf, la, rt.Vec3(0, 1, 0), float(rng.uniform(5, 120)), W / H, float(rng.uniform(0.0, 0.5)) * spread / 10,
                    float(np.linalg.norm(lf - la)) + 1e-3)
    seed = int(rng.integers(1, 2 ** 62))
    u53 = rng.random() < float(os.environ.get("FUZZ_U53", "0"))
    r.upload_scene(flat)
    sm, fix, st = r.render(cam, rt.make_params(W, H, spp, seed=seed, tile_rows=int(rng.integers(1, 9)), flags=rt.RT_FLAG_UNIFORM53 if u53 else 0))
    fb, sb, stb = oracle.render_b(oracle.camera_from_host(cam), flat, oracle.make_params(W, H, spp, seed=seed, uniform53=u53))
    ok = np.array_equal(fix, fb) and st["rays_traced"] == stb["rays_traced"]
    tot_rays += st["rays_traced"]
    if not ok:
        bad += 1
        print(f"MISMATCH case {case}: n={len(flat)} {W}x{H}x{spp} spread={spread:.2f} u53={u53} diff px={int(np.count_nonzero((fix != fb).any(2)))}", flush=True)
    if (case + 1) % 1000 == 0:
        print(f"   ... {case + 1} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print(f"{cases} cases, {bad} mismatches, {tot_rays} rays, mode {os.environ.get('RTIOW_SCAN_MODE', '5 (default)')}, {time.time() - t0:.1f} s", flush=True)
sys.exit(1 if bad else 0)
