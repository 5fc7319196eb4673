"""First-contact GPU check (development tool, not a test): parity of the f32
kernel against oracle B on cfg1, then timing on cfg2, for each scan mode."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
import rtiow_amd as rt

world = rt.random_scene(1)
flat = world.flatten()
print("spheres", len(flat), flush=True)

def parity(mode, W, H, spp):
    os.environ["RTIOW_SCAN_MODE"] = str(mode)
    r = rt.Renderer(0)
    r.upload_scene(flat)
    cam = rt.book1_camera(W, H)
    p = rt.make_params(W, H, spp)
    sm, fix, st = r.render(cam, p)
    c32 = oracle.camera_to_f32(oracle.book1_camera(W, H))
    fb, sb, stb = oracle.render_b(c32, flat, oracle.make_params(W, H, spp))
    nbad = int(np.count_nonzero(fix != fb))
    print(f"mode {mode} {W}x{H}x{spp}: mismatching u64 sums {nbad}/{fix.size}; rays gpu {st['rays_traced']} cpu {stb['rays_traced']}; "
          f"samples {st['samples']}; cand {st['candidates']}; kernel_ms {st['kernel_ms']:.3f}; f32 equal {np.array_equal(sm, sb)}", flush=True)
    if nbad:
        idx = np.argwhere(fix != fb)[:5]
        for i in idx: print("   ", i, fix[tuple(i)], fb[tuple(i)])
    print("   philox", [hex(x) for x in r.philox((0,0,0,0),(0,0))], flush=True)
    r.close()

def timing(mode, W, H, spp, reps=3, chunk=None, bpc=None):
    os.environ["RTIOW_SCAN_MODE"] = str(mode)
    if chunk: os.environ["RTIOW_CHUNK"] = str(chunk)
    if bpc: os.environ["RTIOW_BLOCKS_PER_CU"] = str(bpc)
    r = rt.Renderer(0)
    r.upload_scene(flat)
    cam = rt.book1_camera(W, H)
    p = rt.make_params(W, H, spp)
    for k in range(reps):
        sm, fix, st = r.render(cam, p, want_fix=False)
        ms = st['kernel_ms']
        print(f"mode {mode} chunk {chunk} bpc {bpc} {W}x{H}x{spp}: {ms:.2f} ms  {W*H*spp/ms/1e3:.1f} Msamples/s  rays/sample {st['rays_traced']/st['samples']:.3f} "
              f"cand/ray {st['candidates']/max(1,st['rays_traced']):.2f} grid {st['grid_blocks']}", flush=True)
    r.close()
    os.environ.pop("RTIOW_CHUNK", None); os.environ.pop("RTIOW_BLOCKS_PER_CU", None)

for mode in (0, 1):
    parity(mode, 64, 36, 4)
    parity(mode, 400, 225, 10)
for mode in (0, 1):
    timing(mode, 1200, 675, 100)
timing(1, 1200, 675, 100, chunk=4)
timing(1, 1200, 675, 100, chunk=16)
timing(1, 1200, 675, 100, bpc=2)
timing(1, 1200, 675, 100, bpc=4)
