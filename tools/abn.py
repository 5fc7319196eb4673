"""Diagnostic: median cfg2 kernel time of several builds of the library, interleaved rounds in one process.
usage: python tools/abn.py lib1.so lib2.so ..."""
import ctypes as C, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import numpy as np
from rtiow_amd import _ffi
import rtiow_amd as rt
flat = np.ascontiguousarray(rt.random_scene(1).flatten(), dtype=rt.SPHERE_DTYPE)
cam = rt.book1_camera(1200, 675).to_rt_camera()
p = rt.make_params(1200, 675, int(os.environ.get("SPP", "100")))
libs = []
for path in sys.argv[1:]:
    lib = C.CDLL(os.path.abspath(path))
    for name, res, args in _ffi.SYMBOLS:
        if hasattr(lib, name):
            fn = getattr(lib, name); fn.restype = res; fn.argtypes = args
    h = C.c_void_p()
    assert lib.rt_create(0, C.byref(h)) == 0
    assert lib.rt_upload_scene(h, flat.ctypes.data_as(C.POINTER(_ffi.rt_sphere)), len(flat)) == 0
    libs.append((path, lib, h, []))
out = np.zeros((675, 1200, 3), dtype=np.float32)
st = _ffi.rt_stats()
for rnd in range(7):
    for path, lib, h, ts in libs:
        assert lib.rt_render(h, C.byref(cam), C.byref(p), out.ctypes.data_as(C.c_void_p), None, C.byref(st)) == 0
        if rnd: ts.append(st.kernel_ms)
for path, lib, h, ts in libs:
    print(f"{path}: median {statistics.median(ts):.3f} ms  min {min(ts):.3f}  (n={len(ts)})")
