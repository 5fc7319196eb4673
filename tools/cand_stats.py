"""Diagnostic: candidates and exact roots per ray on the book scene (RT_FLAG_DIAG_STATS)."""
import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtiow_amd as rt
r = rt.Renderer(0)
r.upload_scene(rt.random_scene(1).flatten())
_,_,st = r.render(rt.book1_camera(1200,675), rt.make_params(1200,675,10, flags=rt.RT_FLAG_DIAG_STATS), want_fix=False)
print({k: st[k] for k in st if not k.startswith('rays_per')})
print("cand/ray", st["candidates"]/st["rays_traced"], "roots/ray", st["exact_roots"]/st["rays_traced"] if "exact_roots" in st else None)
