"""Diagnostic (not a test): loads an RT_EXIT_TIMES build (RTIOW_LIB) and prints how far apart the waves of
the persistent grid leave the kernel on cfg2 (SPP=... for another sample count; 100 MHz real-time clock)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from rtiow_amd import _ffi
_ffi.LIB_PATH = os.environ["RTIOW_LIB"]
import rtiow_amd as rt
r = rt.Renderer(0)
r.upload_scene(rt.random_scene(1).flatten())
for _ in range(2):
    sm, fix, st = r.render(rt.book1_camera(1200, 675), rt.make_params(1200, 675, int(os.environ.get("SPP", "100"))), want_fix=False)
out = (C.c_ulonglong * 8)()
r._lib.rt_debug_phase_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
r._lib.rt_debug_phase_cycles(r._h, out)
M = (1 << 64) - 1
last, first_exit, sum_exit, first_start, n = out[0], M - out[1], out[2], M - out[3], out[4]
span = (last - first_start) / 100.0
print(f"kernel {st['kernel_ms']:.2f} ms; {n} waves; first wave starts .. last wave exits: {span:.1f} us")
print(f"   first exit at {100.0 * (first_exit - first_start) / (last - first_start):.1f} % of that span, "
      f"mean exit at {100.0 * (sum_exit / n - first_start) / (last - first_start):.1f} %  "
      f"(idle tail = {100.0 - 100.0 * (sum_exit / n - first_start) / (last - first_start):.1f} % of the wave-time)")
r.close()
