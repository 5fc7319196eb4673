"""Diagnostic (not a test): where the render kernel's LDS bank conflicts come from.  Loads an RT_LDS_CONFLICTS build
(RTIOW_LIB, tools/build_diag_libs.sh) whose kernel models, in software, the bank serialisation of every LDS instruction with a
data-dependent address (rt_kernels.hpp, lds_extra_cycles) and sums the extra cycles per site; prints them per wave-pass next
to the totals the hardware counters give for the same launch with the shipped library (tools/lds_conflicts.sh).
usage: RTIOW_LIB=tools/lib_ldsc.so python tools/lds_conflicts.py [width height spp]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from rtiow_amd import _ffi
_ffi.LIB_PATH = os.environ["RTIOW_LIB"]
import rtiow_amd as rt
names = ["candidate recording ds_or_b32 (same ray from several columns)", "block sums ds_add_u64 (finished samples of one pixel)",
         "pool ds_min_u64 on the per-ray minimum", "pool ds_max_u32 / reset of the per-ray sphere", "pool ds_bpermute of (o, d), 12 per round",
         "enumeration: bitmap word + tile-list reads", "sample queue reads (7 per taken sample)", "pool: 8-byte read of the per-ray minimum"]
w, h, spp = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (600, 338, 100)
big = os.environ.get("SCENE") == "cfg4"
r = rt.Renderer(0)
r.upload_scene(rt.random_scene(1, grid=(-50, 49) if big else (-11, 11)).flatten())
sm, fix, st = r.render(rt.book1_camera(w, h), rt.make_params(w, h, spp), want_fix=False)
out = (C.c_ulonglong * 8)()
r._lib.rt_debug_phase_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
r._lib.rt_debug_phase_cycles(r._h, out)
passes = st["rays_traced"] / 64.0 / 0.978            # (97.8 % of the lane slots of a pass hold a ray)
tot = sum(out)
print(f"{'10k-sphere' if big else 'book'} scene {w}x{h}x{spp}: rays {st['rays_traced']}, ~{passes:.0f} wave-passes; modelled extra LDS cycles (bank conflicts):")
for k in range(8):
    print(f"   {names[k]:70s} {out[k]:12d}   {out[k] / passes:7.2f} per wave-pass   {100.0 * out[k] / max(1, tot):5.1f} %")
print(f"   {'sum':70s} {tot:12d}   {tot / passes:7.2f} per wave-pass")
r.close()
