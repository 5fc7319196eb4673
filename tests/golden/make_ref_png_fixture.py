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


# ---- second fixture: what the PNG holds about the three BIG spheres ---------------------------------------
# The three big spheres are hard-coded (main.rs:93-99) and so is the camera (main.rs:108-118): everything
# about them that does not involve the randomly placed small spheres is a known answer too.
#  (i)   the fuzz-0 Metal sphere at (4,1,0), albedo (0.7,0.6,0.5): where its mirror image is the sky, a pixel
#        is albedo (x) sky(reflected direction) -- pins Sphere::hit (root, point, normal), HitRecord::new,
#        Metal::scatter, Vec3::reflect and the attenuation product of ray_color.  Mask, from geometry only
#        (pinhole ray through the pixel centre): the ray hits that sphere with -d.n > 0.3 (off the rim), the
#        mirrored direction rises (unit y > 0.12) and clears the other two big spheres.
#  (ii)  the horizon of the radius-1000 ground sphere: first non-sky row per column, left and right of the big
#        spheres (far small spheres can touch it: compared statistically).
#  (iii) silhouettes above the horizon: leftmost / rightmost non-sky pixel per row (Lambertian sphere at
#        (-4,1,0), Metal sphere at (4,1,0)), topmost non-sky pixel per column (all three).
#  (iv)  the mean colour of a patch of the Lambertian sphere (albedo (0.4,0.2,0.1)) that faces the sky.
#  (v)   the mean colour of a patch in the LOWER half of the Dialectric sphere at (0,1,0), ir 1.5: refraction turns
#        the image upside down, so that half shows the sky -- camera ray refracted in, refracted out
#        (materials.rs:76-105, vec3.rs:120-125), minus the few per cent Schlick sends elsewhere.  The patch mean does
#        not depend on the small spheres (three scene seeds agree within 0.3) but moves by 1.1 for ir 1.45 and 2.1
#        for ir 1.6.
# "non-sky" = differs from the sky the camera would see there by more than 6 in some channel; the sky
# reference is Oracle A on an EMPTY scene (this script is test infrastructure and may load the oracle).
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import oracle  # noqa: E402

W, H = 1200, 800
DST2 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_png_spheres.json")
cam = oracle.camera_new((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, 3.0 / 2.0, 0.1, 10.0)


def v(a):
    return np.array([a[0], a[1], a[2]], dtype=np.float64)


def metal_cap_mask(y):
    """x -> True where the pixel centre's pinhole ray sees sky mirrored in the metal sphere (see (i))."""
    org, llc, hor, ver = v(cam.origin), v(cam.lower_left_corner), v(cam.horizontal), v(cam.vertical)
    j = H - 1 - y
    x = np.arange(W)
    d = llc[None, :] + ((x + 0.5) / (W - 1))[:, None] * hor[None, :] + ((j + 0.5) / (H - 1)) * ver[None, :] - org[None, :]
    d /= np.linalg.norm(d, axis=1, keepdims=True)

    def hit(o, dd, c, r):
        oc = o - c
        hb = (oc * dd).sum(-1)
        disc = hb * hb - ((oc * oc).sum(-1) - r * r)
        t = -hb - np.sqrt(np.maximum(disc, 0.0))
        return (disc > 0) & (t > 1e-4), t

    ok, t = hit(org[None, :], d, np.array([4.0, 1.0, 0.0]), 1.0)
    p = org[None, :] + t[:, None] * d
    n = p - np.array([4.0, 1.0, 0.0])
    cosv = -(d * n).sum(-1)
    refl = d - 2.0 * (d * n).sum(-1, keepdims=True) * n
    ok &= (cosv > 0.3) & (refl[:, 1] > 0.12)
    for c, r in (((0.0, 1.0, 0.0), 1.0), ((-4.0, 1.0, 0.0), 1.0), ((0.0, -1000.0, 0.0), 1000.0)):
        blocked, _ = hit(p, refl, np.array(c), r)
        ok &= ~blocked
    return ok


def sky_rows(y0, y1):
    p = oracle.make_params(W, H, 4, rows=(H - y1, H - y0, 1))
    s, _ = oracle.render_a(cam, np.zeros(0, dtype=np.dtype([("b", "V72")])), p)
    return oracle.resolve_a(s, 4, flip=True)[:, :, :3].astype(int)


cap = {}
for y in (90, 120, 160, 200, 240, 270, 285):
    m = metal_cap_mask(y)
    xs = np.nonzero(m)[0]
    runs = np.split(xs, np.nonzero(np.diff(xs) != 1)[0] + 1)      # the longest run of the row
    xs = max(runs, key=len)
    assert len(xs) > 150, y
    cap[str(y)] = {"x0": int(xs[0]), "x1": int(xs[-1]) + 1, "rgb": im[y, xs[0]:xs[-1] + 1, :3].astype(int).tolist()}

Y0, Y1 = 40, 230
sky = sky_rows(Y0, Y1)
nonsky = np.abs(im[Y0:Y1, :, :3].astype(int) - sky).max(axis=2) > 6
hor_cols = [x for x in range(0, W, 8) if x < 330 or x > 1070]
horizon = {str(x): int(Y0 + np.argmax(nonsky[:, x])) for x in hor_cols}
left_right = {}
for y in range(60, 180, 10):
    xs = np.nonzero(nonsky[y - Y0])[0]
    left_right[str(y)] = [int(xs.min()), int(xs.max())]
top = {str(x): int(Y0 + np.argmax(nonsky[:, x])) for x in range(460, 1001, 20)}
patch = {"x0": 380, "x1": 440, "y0": 110, "y1": 170}
patch["mean_rgb"] = [float(c) for c in im[patch["y0"]:patch["y1"], patch["x0"]:patch["x1"], :3].reshape(-1, 3).mean(0)]
glass = {"x0": 470, "x1": 540, "y0": 260, "y1": 320}
glass["mean_rgb"] = [float(c) for c in im[glass["y0"]:glass["y1"], glass["x0"]:glass["x1"], :3].reshape(-1, 3).mean(0)]
json.dump({"source": "rtiow_part1_final.png", "width": W, "height": H, "nonsky_threshold": 6, "rows_scanned": [Y0, Y1],
           "dialectric_patch": glass,
           "metal_cap_rows": cap, "horizon_first_nonsky_row": horizon, "silhouette_left_right": left_right,
           "silhouette_top": top, "lambertian_patch": patch}, open(DST2, "w"))
print("wrote", DST2)
