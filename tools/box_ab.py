"""Diagnostic: kernel time of the sky rows / ground rows / whole frame of cfg2 with and without the scene-box cull."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtiow_amd as rt
flat = rt.random_scene(1).flatten()
w, h, spp = 1200, 675, 100
cam = rt.book1_camera(w, h)
for nocull in ("0", "1"):
    os.environ["RTIOW_NO_BOX_CULL"] = nocull
    r = rt.Renderer(0)
    r.upload_scene(flat)
    for name, kw in (("whole frame", dict()), ("rows 540..674 (sky)", dict(tile_rows=135, shard_index=4, shard_count=5)),
                     ("rows 0..134 (ground)", dict(tile_rows=135, shard_index=0, shard_count=5))):
        ms = []
        for _ in range(4):
            _, _, st = r.render(cam, rt.make_params(w, h, spp, **kw), want_fix=False)
            ms.append(st["kernel_ms"])
        print(f"no_cull={nocull} {name:24s} kernel {min(ms):8.3f} ms  rays/sample {st['rays_traced']/st['samples']:.3f}  samples {st['samples']}")
    r.close()
