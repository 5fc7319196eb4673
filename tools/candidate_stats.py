"""Diagnostic: pooled candidates and computed roots per ray (RT_FLAG_DIAG_STATS) on the book scene and the 10k-sphere scene."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtiow_amd as rt
r = rt.Renderer(0)
for name, grid, w, h, spp in (("book", (-11, 11), 1200, 675, 20), ("10k", (-50, 49), 1920, 1080, 8)):
    r.upload_scene(rt.random_scene(1, grid=grid).flatten())
    _, _, st = r.render(rt.book1_camera(w, h), rt.make_params(w, h, spp, flags=rt.RT_FLAG_DIAG_STATS), want_fix=False)
    rays = st["rays_traced"]
    print(name, "rays", rays, "candidates/ray %.3f" % (st["candidates"] / rays), "roots/ray %.3f" % (st["exact_roots"] / rays), "sphere_tests/ray %.2f" % (st["sphere_tests"] / rays))
