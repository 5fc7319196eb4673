"""A/B two builds of the library in ONE process on ONE device, interleaved rounds (cfg2)."""
import ctypes as C, os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# each library needs its own process-level handle: run the measurement in child processes would mix
# devices; instead load both .so files side by side through ctypes with distinct handles
import torch  # noqa
import numpy as np
from rtiow_amd import _ffi
import rtiow_amd as rt
libs = sys.argv[1:]
flat = rt.random_scene(1).flatten()
handles = []
for path in libs:
    lib = C.CDLL(os.path.abspath(path))
    for name, res, args in _ffi.SYMBOLS:
        if hasattr(lib, name):
            fn = getattr(lib, name); fn.restype = res; fn.argtypes = args
    h = C.c_void_p(); assert lib.rt_create(0, C.byref(h)) == 0
    ptr = flat.ctypes.data_as(C.POINTER(_ffi.rt_sphere)); assert lib.rt_upload_scene(h, ptr, len(flat)) == 0
    handles.append((path, lib, h))
cam = rt.book1_camera(1200, 675).to_rt_camera(); p = rt.make_params(1200, 675, 100)
out = np.zeros((675, 1200, 3), np.float32); st = _ffi.rt_stats()
times = {path: [] for path, _, _ in handles}
for rnd in range(8):
    for path, lib, h in handles:
        assert lib.rt_render(h, C.byref(cam), C.byref(p), out.ctypes.data_as(C.c_void_p), None, C.byref(st)) == 0
        if rnd >= 2: times[path].append(st.kernel_ms)
for path in times:
    t = np.array(times[path]); print(f"{path}: median {np.median(t):.3f} ms  min {t.min():.3f}  (n={len(t)})")
