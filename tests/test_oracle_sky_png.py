"""Oracle A against the reference's own committed render.

rtiow_part1_final.png (1200x800, 3:2) was rendered from an OS-seeded scene, so
only its SKY pixels are a known answer: they depend on Camera::new
(camera.rs:17-45), the direction of Camera::get_ray (camera.rs:47-54), the sky
branch of ray_color (main.rs:54-56), Color::to_rgba (vec3.rs:403-421) and the
row flip (main.rs:141-145) -- and on nothing random beyond sub-pixel jitter.
The fixture (tests/golden/ref_png_sky_rows.json) was extracted by
tests/golden/make_ref_png_fixture.py.
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

W, H = 1200, 800


@pytest.fixture(scope="module")
def fx():
    return json.load(open(os.path.join(GOLDEN, "ref_png_sky_rows.json")))


def _render_row(oracle_mod, flat, y, spp):
    cam = oracle_mod.camera_new((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, 3.0 / 2.0, 0.1, 10.0)   # main.rs:108-118
    j = H - 1 - y                                                 # flip, main.rs:141-145
    p = oracle_mod.make_params(W, H, spp, rows=(j, j + 1, 1))
    sums, _ = oracle_mod.render_a(cam, flat, p)
    return oracle_mod.resolve_a(sums, spp, flip=False)[0]


# rows at least 7 rows away from a u8 boundary of the gradient (the PNG flips 220->221 at rows 15-16)
@pytest.mark.parametrize("y", [0, 1, 4, 8, 24, 37, 50])
def test_constant_sky_rows_match_png_exactly(oracle_mod, book1_flat, fx, y):
    want = np.array(fx["constant_rows"][str(y)], dtype=np.uint8)
    got = _render_row(oracle_mod, book1_flat, y, spp=32)
    assert (got[:, 3] == 255).all()
    assert (got[:, :3] == want).all()


def test_boundary_rows_only_take_the_two_png_values(oracle_mod, book1_flat, fx):
    for y, allowed in fx["boundary_rows"].items():
        got = _render_row(oracle_mod, book1_flat, int(y), spp=8)[:, :3]
        assert all(list(px) in allowed for px in np.unique(got, axis=0).tolist())


def test_lower_sky_points_match_png(oracle_mod, book1_flat, fx):
    for pt in fx["points"]:
        if pt["x"] == 600:
            continue                       # covered by a sphere in the PNG's (unknown) scene
        got = _render_row(oracle_mod, book1_flat, pt["y"], spp=16)[pt["x"], :3]
        assert got.tolist() == pt["rgb"], pt
