"""What the reference's committed render (rtiow_part1_final.png, 1200x800) holds about the three hard-coded
big spheres (main.rs:93-99) under the hard-coded camera (main.rs:108-118), as checks that take any renderer
of the book-1 scene (the CPU oracle; the GPU path).  Fixture: tests/golden/ref_png_spheres.json, extracted
by tests/golden/make_ref_png_fixture.py (which documents each item).  The scene used here (random_scene seed 1)
has OTHER small spheres than the PNG's OS-seeded scene, so only content that does not depend on the small
spheres is compared."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
W, H = 1200, 800


def fixture():
    return json.load(open(os.path.join(GOLDEN, "ref_png_spheres.json")))


def check_metal_cap(fx, row_rgb):
    """row_rgb(y) -> [W,3] u8 of image row y (0 = top).  Where the fuzz-0 Metal sphere mirrors the sky the pixel is
    albedo (x) sky(reflect(d, n)): within 1 of the PNG on the geometric mask (defocus and jitter only blur a
    smooth gradient).  Pins sphere.rs:15-41, mod.rs:20-30, materials.rs:48-62, vec3.rs:116-118, main.rs:49."""
    n = 0
    for y, rec in fx["metal_cap_rows"].items():
        got = row_rgb(int(y))[rec["x0"]:rec["x1"]].astype(int)
        want = np.array(rec["rgb"], dtype=int)
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= 1, f"row {y}: max |diff| {np.abs(got - want).max()}"
        n += len(want)
    assert n > 2000


def nonsky_mask(fx, rows_rgb, sky_rgb):
    """rows_rgb, sky_rgb: [y1-y0, W, 3] of rows fx['rows_scanned'] of the scene and of the EMPTY scene."""
    return np.abs(rows_rgb.astype(int) - sky_rgb.astype(int)).max(axis=2) > fx["nonsky_threshold"]


def check_silhouettes_and_horizon(fx, nonsky):
    y0 = fx["rows_scanned"][0]
    # (ii) horizon of the ground sphere: far small spheres can touch it, so statistically
    d = np.array([(y0 + int(np.argmax(nonsky[:, int(x)]))) - want for x, want in fx["horizon_first_nonsky_row"].items()])
    assert np.median(d) == 0 and np.mean(np.abs(d) <= 1) >= 0.8 and np.abs(d).max() <= 3, d
    # (iii) left edge = the Lambertian sphere at (-4,1,0) (out of focus: +-3 px), right edge = the Metal sphere at (4,1,0)
    for y, (left, right) in fx["silhouette_left_right"].items():
        xs = np.nonzero(nonsky[int(y) - y0])[0]
        assert abs(int(xs.min()) - left) <= 3 and abs(int(xs.max()) - right) <= 2, (y, xs.min(), xs.max(), left, right)
    # top edge per column: Lambertian, Dialectric (0,1,0) and Metal spheres
    for x, want in fx["silhouette_top"].items():
        got = y0 + int(np.argmax(nonsky[:, int(x)]))
        assert abs(got - want) <= 2, (x, got, want)


def check_lambertian_patch(fx, rows_rgb):
    """Mean colour of a sky-facing patch of the Lambertian sphere (albedo (0.4,0.2,0.1)): the cosine-weighted
    scatter of materials.rs:21-31 (normal + random_unit_vector) and the albedo product; a uniform-in-sphere
    scatter or a missing normalisation shifts it by several units."""
    p, y0 = fx["lambertian_patch"], fx["rows_scanned"][0]
    got = rows_rgb[p["y0"] - y0:p["y1"] - y0, p["x0"]:p["x1"]].reshape(-1, 3).astype(float).mean(0)
    assert np.abs(got - np.array(p["mean_rgb"])).max() <= 1.5, (got, p["mean_rgb"])


def check_dialectric_patch(fx, rows_rgb_of):
    """rows_rgb_of(y0, y1) -> [y1-y0, W, 3].  Mean colour of a patch in the lower half of the Dialectric sphere
    (0,1,0), ir 1.5, which shows the refracted (upside-down) sky: pins refract(), the front/back ratio, total
    internal reflection and the Schlick mix of materials.rs:76-105 statistically -- the mean moves by 1.1 for
    ir 1.45 and by 2.1 for ir 1.6, and by < 0.3 between differently seeded small spheres."""
    p = fx["dialectric_patch"]
    got = rows_rgb_of(p["y0"], p["y1"])[:, p["x0"]:p["x1"]].reshape(-1, 3).astype(float).mean(0)
    assert np.abs(got - np.array(p["mean_rgb"])).max() <= 0.8, (got, p["mean_rgb"])
