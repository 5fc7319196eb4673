"""Diagnostic: candidate statistics and timing per scan mode on cfg2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from rtiow_amd import _ffi
if os.environ.get('RTIOW_LIB'): _ffi.LIB_PATH = os.environ['RTIOW_LIB']
import rtiow_amd as rt
flat = rt.random_scene(1).flatten()
for mode in [int(x) for x in os.environ.get('MODES', '3,2,1').split(',')]:
    for chunk in [int(x) for x in os.environ.get('CHUNKS', '4').split(',')]:
      for ib in [int(x) for x in os.environ.get('IBS', '256').split(',')]:
        os.environ['RTIOW_ITEM_BLOCK'] = str(ib)
        os.environ["RTIOW_SCAN_MODE"] = str(mode); os.environ["RTIOW_CHUNK"] = str(chunk)
        r = rt.Renderer(0); r.upload_scene(flat)
        for _ in range(2):
            sm, fix, st = r.render(rt.book1_camera(1200, 675), rt.make_params(1200, 675, 100, flags=rt.RT_FLAG_DIAG_STATS if os.environ.get('DIAG') else 0), want_fix=False)
        print(f"mode {mode} chunk {chunk} item_block {ib}: {st['kernel_ms']:.2f} ms cand/ray {st['candidates']/st['rays_traced']:.3f} roots/ray {st['exact_roots']/st['rays_traced']:.3f} grid {st['grid_blocks']}", flush=True)
        r.close()
