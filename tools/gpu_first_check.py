"""Development tool (not a test): parity of the kernel against oracle B, the
filtered vs unfiltered scan, f64 div/sqrt rounding, and timing on cfg2."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
import rtiow_amd as rt

flat = rt.random_scene(1).flatten()
print("spheres", len(flat), flush=True)

def parity(W, H, spp, flags=0):
    r = rt.Renderer(0)
    r.upload_scene(flat)
    cam = rt.book1_camera(W, H)
    p = rt.make_params(W, H, spp, flags=flags)
    sm, fix, st = r.render(cam, p)
    fb, sb, stb = oracle.render_b(oracle.camera_from_host(cam), flat, oracle.make_params(W, H, spp))
    nbad = int(np.count_nonzero(fix != fb))
    print(f"flags {flags} {W}x{H}x{spp}: mismatching u64 sums {nbad}/{fix.size}; rays gpu {st['rays_traced']} cpu {stb['rays_traced']}; "
          f"samples {st['samples']}; cand/ray {st['candidates']/st['rays_traced']:.2f} roots/ray {st['exact_roots']/st['rays_traced']:.2f}; "
          f"kernel_ms {st['kernel_ms']:.3f}; f32 equal {np.array_equal(sm, sb)}", flush=True)
    if nbad:
        idx = np.argwhere(fix != fb)[:5]
        for i in idx: print("   ", i, fix[tuple(i)], fb[tuple(i)])
    rg = r.resolve_rgba8(fix, spp); rb = oracle.resolve_b(fb, spp)
    print("   rgba equal", np.array_equal(rg, rb), flush=True)
    r.close()

def kat():
    r = rt.Renderer(0)
    rng = np.random.default_rng(0)
    a = np.abs(rng.standard_normal(1 << 20)) * 10.0 ** rng.integers(-20, 20, 1 << 20)
    b = rng.standard_normal(1 << 20) * 10.0 ** rng.integers(-20, 20, 1 << 20)
    q, s = r.f64_div_sqrt(a, b)
    print("f64 div mismatches", int(np.count_nonzero(q != a / b)), "sqrt mismatches", int(np.count_nonzero(s != np.sqrt(a))), flush=True)
    print("philox", [hex(x) for x in r.philox((0,0,0,0),(0,0))], flush=True)
    r.close()

def timing(W, H, spp, reps=3, chunk=None, bpc=None, flags=0):
    if chunk: os.environ["RTIOW_CHUNK"] = str(chunk)
    if bpc: os.environ["RTIOW_BLOCKS_PER_CU"] = str(bpc)
    r = rt.Renderer(0)
    r.upload_scene(flat)
    cam = rt.book1_camera(W, H)
    p = rt.make_params(W, H, spp, flags=flags)
    for k in range(reps):
        sm, fix, st = r.render(cam, p, want_fix=False)
        ms = st['kernel_ms']
        print(f"flags {flags} chunk {chunk} bpc {bpc} {W}x{H}x{spp}: {ms:.2f} ms  {W*H*spp/ms/1e3:.1f} Msamples/s  rays/sample {st['rays_traced']/st['samples']:.3f} "
              f"cand/ray {st['candidates']/max(1,st['rays_traced']):.2f} roots/ray {st['exact_roots']/max(1,st['rays_traced']):.2f} grid {st['grid_blocks']}", flush=True)
    r.close()
    os.environ.pop("RTIOW_CHUNK", None); os.environ.pop("RTIOW_BLOCKS_PER_CU", None)

kat()
parity(64, 36, 4)
parity(64, 36, 4, flags=rt.RT_FLAG_NO_FILTER)
parity(400, 225, 10)
parity(400, 225, 10, flags=rt.RT_FLAG_NO_FILTER)
timing(1200, 675, 100)
for ch in (1, 2, 4):
    timing(1200, 675, 100, chunk=ch, reps=2)
timing(1200, 675, 100, bpc=3, reps=2)
timing(1200, 675, 100, bpc=4, reps=2)
timing(400, 225, 10, flags=rt.RT_FLAG_NO_FILTER, reps=1)
