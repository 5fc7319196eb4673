"""Extracts the deterministic sky pixels of the reference's committed render
(/root/reference/rtiow_part1_final.png, 1200x800 RGBA8) into a small JSON
fixture.  The render's scene was OS-seeded, so only its sky is a known answer
(SURVEY.md section 4): it pins Camera::new, Camera::get_ray's direction, the sky
gradient of ray_color, Color::to_rgba and the row flip.

Run once in the build container (the reference does not travel):
    python tests/golden/make_ref_png_fixture.py
"""
import json
import os

import numpy as np
from PIL import Image

SRC = "/root/reference/rtiow_part1_final.png"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_png_sky_rows.json")

im = np.array(Image.open(SRC))
assert im.shape == (800, 1200, 4) and (im[:, :, 3] == 255).all()
rows, mixed = {}, {}
for y in range(0, 51):                      # image rows 0..50 (top) are pure sky
    row = im[y, :, :3]
    if (row == row[0]).all():               # constant across x
        rows[str(y)] = [int(v) for v in row[0]]
    else:                                   # a u8 quantisation boundary falls inside the row:
        vals = np.unique(row.reshape(-1, 3), axis=0)   # jitter decides which side a pixel lands on
        assert len(vals) == 2 and np.abs(vals[0].astype(int) - vals[1].astype(int)).sum() == 1, y
        mixed[str(y)] = [[int(v) for v in c] for c in vals]
points = []
for y in (100, 150):
    for x in (0, 600, 1199):
        points.append({"y": y, "x": x, "rgb": [int(v) for v in im[y, x, :3]]})
json.dump({"source": "rtiow_part1_final.png", "width": 1200, "height": 800,
           "aspect": [3, 2], "constant_rows": rows, "boundary_rows": mixed, "points": points,
           "image_mean_rgb": [float(v) for v in im[:, :, :3].reshape(-1, 3).mean(0)]},
          open(DST, "w"), indent=1)
print("wrote", DST)
