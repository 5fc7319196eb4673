"""Diagnostic (not a test): loads the RT_PHASE_STAMPS build and prints the share of
wave time per phase of the bounce loop on cfg2."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from rtiow_amd import _ffi
_ffi.LIB_PATH = os.environ.get("RTIOW_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib_stamps.so")   # tools/build_diag_libs.sh
import rtiow_amd as rt
names = ["(a) take samples from the queue", "-", "(d) pooled exact rounds", "(d) tile list", "(f) quantize + accumulate", "(d) operands + always-exact", "(d) matrix tile loop", "(d) footprints",
         "(b) refill: start 64 samples", "(a) camera ray of fresh lanes", "(d) enumerate candidates", "(e) hit record + first Philox block", "(e) unit-sphere retry loop", "(e) unit_vector + materials", "-", "-"]
for mode in [int(x) for x in os.environ.get("MODES", "4").split(",")]:
    os.environ["RTIOW_SCAN_MODE"] = str(mode)
    r = rt.Renderer(0)
    big = os.environ.get("SCENE") == "cfg4"                      # the 10k-sphere scene instead of the book scene
    r.upload_scene(rt.random_scene(1, grid=(-50, 49) if big else (-11, 11)).flatten())
    w, h, spp = (1920, 1080, 32) if big else (1200, 675, 100)
    spp = int(os.environ.get("SPP", spp))
    for _ in range(2):
        sm, fix, st = r.render(rt.book1_camera(w, h), rt.make_params(w, h, spp), want_fix=False)
    out = (C.c_ulonglong * 16)()
    r._lib.rt_debug_phase_cycles16.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    r._lib.rt_debug_phase_cycles16(r._h, out)
    tot = sum(out[:16])
    print(f"scan mode {mode}: kernel {st['kernel_ms']:.2f} ms (stamped build), wave-time shares:")
    for k in range(16):
        print(f"   {names[k]:38s} {100.0 * out[k] / tot:6.2f} %   {out[k] / max(1, st['rays_traced'] / 64):10.0f} ticks per wave-iteration")
    r.close()
