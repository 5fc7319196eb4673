"""Renders the book-1 final scene on cuda:0 through the C ABI and writes it as PNG (visual artefact).
usage: python tools/render_png.py out.png [width height spp]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
import rtiow_amd as rt
out = sys.argv[1]
w, h, spp = (int(x) for x in sys.argv[2:5]) if len(sys.argv) >= 5 else (1200, 675, 100)
with rt.Renderer(0) as r:
    r.upload_scene(rt.random_scene(1))
    sums, fix, st = r.render(rt.book1_camera(w, h), rt.make_params(w, h, spp))
    rgba = r.resolve_rgba8(fix, spp, flip=True)
Image.fromarray(np.ascontiguousarray(rgba[..., :3]), "RGB").save(out, optimize=True)
print(f"{out}: {w}x{h}x{spp} spp, {st['kernel_ms']:.2f} ms kernel, {st['samples'] / st['kernel_ms'] / 1e3:.0f} Msamples/s, "
      f"mean rgb {rgba[..., :3].reshape(-1, 3).mean(0).round(2).tolist()}")
