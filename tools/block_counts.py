"""Diagnostic (not a test): loads an RT_BLOCK_COUNTS build (RTIOW_LIB) and prints how often each
main block of the bounce loop executes per wave pass on cfg2."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from rtiow_amd import _ffi
_ffi.LIB_PATH = os.environ["RTIOW_LIB"]
import rtiow_amd as rt
names = ["bounce-loop passes", "camera block", "16-ray group x tile: MFMA + look", "keep path (half-looks with a hit)",
         "keep path ray-group entries", "bitmap walk trips", "unit-sphere tries (wave level)", "tile iterations"]
r = rt.Renderer(0)
big = os.environ.get("SCENE") == "cfg4"
r.upload_scene(rt.random_scene(1, grid=(-50, 49) if big else (-11, 11)).flatten())
w, h, spp = (1920, 1080, 32) if big else (1200, 675, 100)
spp = int(os.environ.get("SPP", spp))
sm, fix, st = r.render(rt.book1_camera(w, h), rt.make_params(w, h, spp), want_fix=False)
out = (C.c_ulonglong * 8)()
r._lib.rt_debug_phase_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
r._lib.rt_debug_phase_cycles(r._h, out)
print(f"kernel {st['kernel_ms']:.2f} ms, rays {st['rays_traced']}, rays/64 = {st['rays_traced'] / 64:.0f}")
for k in range(8):
    print(f"   {names[k]:36s} {out[k]:12d}   {out[k] / max(1, out[0]):8.3f} per pass")
r.close()
