"""Oracle A against what the reference's committed render holds about the three hard-coded big spheres
(tests/png_pins.py): with test_oracle_sky_png.py this pins, on the reference's own artefact, Sphere::hit,
HitRecord::new, Metal::scatter + reflect, the Lambertian scatter distribution and the sphere/camera geometry --
SURVEY.md section 8 rows a5-a10 (the Dialectric through the mean of the refracted sky in its lower half; what the
PNG cannot pin is the equal-t tie rule and t_min strictness: analytic known answers only)."""
import numpy as np
import pytest

import png_pins
from png_pins import W, H


@pytest.fixture(scope="module")
def fx():
    return png_pins.fixture()


@pytest.fixture(scope="module")
def cam(oracle_mod):
    return oracle_mod.camera_new((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, 3.0 / 2.0, 0.1, 10.0)     # main.rs:108-118


def render_rows(oracle_mod, cam, flat, y0, y1, spp):
    p = oracle_mod.make_params(W, H, spp, rows=(H - y1, H - y0, 1))
    sums, _ = oracle_mod.render_a(cam, flat, p)
    return oracle_mod.resolve_a(sums, spp, flip=True)[:, :, :3]          # first row = image row y0


@pytest.fixture(scope="module")
def scanned(oracle_mod, cam, book1_flat, fx):
    y0, y1 = fx["rows_scanned"]
    rows = render_rows(oracle_mod, cam, book1_flat, y0, y1, 32)
    sky = render_rows(oracle_mod, cam, book1_flat[:0], y0, y1, 4)
    return rows, sky


def test_metal_sphere_mirrors_the_sky_as_in_the_png(oracle_mod, cam, book1_flat, fx):
    png_pins.check_metal_cap(fx, lambda y: render_rows(oracle_mod, cam, book1_flat, y, y + 1, 48)[0])


def test_silhouettes_and_horizon_as_in_the_png(fx, scanned):
    rows, sky = scanned
    png_pins.check_silhouettes_and_horizon(fx, png_pins.nonsky_mask(fx, rows, sky))


def test_lambertian_patch_mean_as_in_the_png(fx, scanned):
    png_pins.check_lambertian_patch(fx, scanned[0])


def test_glass_sphere_shows_the_refracted_sky_as_in_the_png(oracle_mod, cam, book1_flat, fx):
    png_pins.check_dialectric_patch(fx, lambda y0, y1: render_rows(oracle_mod, cam, book1_flat, y0, y1, 64))
    other = book1_flat.copy()                              # ir 1.6 instead of 1.5 must not pass
    k = int(np.nonzero((other["center"] == (0.0, 1.0, 0.0)).all(axis=1))[0][0])
    other["param"][k] = 1.6
    with pytest.raises(AssertionError):
        png_pins.check_dialectric_patch(fx, lambda y0, y1: render_rows(oracle_mod, cam, other, y0, y1, 64))


def test_the_checks_can_fail(oracle_mod, cam, book1_flat, fx):
    """A metal sphere moved by a quarter of its radius, or with 4 % more blue in its albedo, must not pass
    (the pins are not vacuous)."""
    moved = book1_flat.copy()
    k = int(np.nonzero((moved["center"] == (4.0, 1.0, 0.0)).all(axis=1))[0][0])
    moved["center"][k, 1] += 0.25
    with pytest.raises(AssertionError):
        png_pins.check_metal_cap(fx, lambda y: render_rows(oracle_mod, cam, moved, y, y + 1, 16)[0])
    tinted = book1_flat.copy()
    tinted["albedo"][k] = (0.7, 0.6, 0.52)
    with pytest.raises(AssertionError):
        png_pins.check_metal_cap(fx, lambda y: render_rows(oracle_mod, cam, tinted, y, y + 1, 16)[0])
