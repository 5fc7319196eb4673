"""Diagnostic: median cfg2 kernel time for a list of environment settings (each a fresh context).
usage: python tools/env_sweep.py "RTIOW_CHUNK=2" "RTIOW_CHUNK=4 RTIOW_TAIL_SPP=16" ..."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import rtiow_amd as rt
flat = rt.random_scene(1).flatten()
cam = rt.book1_camera(1200, 675)
base = dict(os.environ)
for rnd in range(2):
    for spec in sys.argv[1:]:
        os.environ.clear(); os.environ.update(base)
        for kv in spec.split():
            k, v = kv.split("="); os.environ[k] = v
        r = rt.Renderer(0); r.upload_scene(flat)
        ts = []
        for _ in range(6):
            sm, fix, st = r.render(cam, rt.make_params(1200, 675, 100), want_fix=False)
            ts.append(st["kernel_ms"])
        print(f"[{spec}] median {statistics.median(ts[1:]):.3f} ms  min {min(ts):.3f}", flush=True)
        r.close()
