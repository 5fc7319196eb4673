"""Host mirror of the reference's scene types and of random_scene.

Same names as the reference (including its spelling `Dialectric`):
  Camera             src/camera.rs:4-45   (new; get_ray runs on the device)
  Sphere             src/shapes/sphere.rs:9-13,44-52
  Lambertian/Metal/Dialectric   src/materials.rs:9-19,34-46,64-74
  HittableList       src/shapes/mod.rs:52 (Vec<Box<dyn Hit>>; push keeps order)
  random_scene       src/main.rs:59-102

The reference keeps sphere and material fields private behind trait objects,
so a GPU backend cannot read them back; here every object also knows how to
flatten itself into the 72-byte (all-f64) rt_sphere record of include/rtiow_hip.h.
"""
import math

import numpy as np

from . import _ffi
from .philox import UniformStream

RT_LAMBERTIAN, RT_METAL, RT_DIALECTRIC = 0, 1, 2

SPHERE_DTYPE = np.dtype([("center", "<f8", 3), ("radius", "<f8"), ("albedo", "<f8", 3),
                         ("param", "<f8"), ("kind", "<i4"), ("reserved", "<i4")])
assert SPHERE_DTYPE.itemsize == 72


def Vec3(x, y, z):
    return np.array([x, y, z], dtype=np.float64)


Point3 = Vec3
Color = Vec3


def _unit(v):
    # vec3.rs:107-109 with Div<f64> = multiply by the reciprocal (:371-375)
    return v * (1.0 / math.sqrt(float(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])))


def _cross(a, b):
    return Vec3(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


class Camera:
    """camera.rs:17-45, evaluated in f64 on the host."""

    def __init__(self, look_from, look_at, v_up, v_fov, aspect_ratio, aperture, focus_dist):
        look_from = np.asarray(look_from, dtype=np.float64)
        look_at = np.asarray(look_at, dtype=np.float64)
        v_up = np.asarray(v_up, dtype=np.float64)
        theta = v_fov * (math.pi / 180.0)                # f64::to_radians
        viewport_height = 2.0 * math.tan(theta / 2.0)
        viewport_width = aspect_ratio * viewport_height
        w = _unit(look_from - look_at)
        u = _unit(_cross(v_up, w))
        v = _cross(w, u)
        self.origin = look_from
        self.horizontal = (focus_dist * viewport_width) * u
        self.vertical = (focus_dist * viewport_height) * v
        self.lower_left_corner = ((look_from - self.horizontal * (1.0 / 2.0))
                                  - self.vertical * (1.0 / 2.0)) - focus_dist * w
        self.u, self.v, self.w = u, v, w
        self.lens_radius = aperture / 2.0

    def to_rt_camera(self):
        c = _ffi.rt_camera()
        for name in ("origin", "lower_left_corner", "horizontal", "vertical", "u", "v"):
            a = getattr(self, name)
            setattr(c, name, (_ffi.C.c_double * 3)(float(a[0]), float(a[1]), float(a[2])))
        c.lens_radius = float(self.lens_radius)
        return c


class Scatter:
    """materials.rs:5-7 (the trait); subclasses carry the parameters only --
    scatter() itself runs in the HIP kernel."""
    kind = -1

    def flat(self):
        raise NotImplementedError


class Lambertian(Scatter):
    kind = RT_LAMBERTIAN

    def __init__(self, a):
        self.albedo = np.asarray(a, dtype=np.float64)

    def flat(self):
        return self.kind, self.albedo, 0.0


class Metal(Scatter):
    kind = RT_METAL

    def __init__(self, a, fuzz):
        self.albedo = np.asarray(a, dtype=np.float64)
        self.fuzz = float(fuzz)

    def flat(self):
        return self.kind, self.albedo, self.fuzz


class Dialectric(Scatter):
    kind = RT_DIALECTRIC

    def __init__(self, index_of_refraction):
        self.ir = float(index_of_refraction)

    def flat(self):
        return self.kind, np.zeros(3), self.ir


class Sphere:
    """shapes/sphere.rs:44-52."""

    def __init__(self, cen, r, mat):
        self.center = np.asarray(cen, dtype=np.float64)
        self.radius = float(r)
        self.mat = mat


class HittableList(list):
    """shapes/mod.rs:52.  `push` mirrors Vec::push; order is part of the input."""

    def push(self, obj):
        self.append(obj)

    def flatten(self):
        out = np.zeros(len(self), dtype=SPHERE_DTYPE)
        for i, s in enumerate(self):
            kind, albedo, param = s.mat.flat()
            out[i]["center"] = s.center
            out[i]["radius"] = s.radius
            out[i]["kind"] = kind
            out[i]["albedo"] = albedo
            out[i]["param"] = param
        return out


def random_scene(seed=1, grid=(-11, 11)):
    """main.rs:59-102 with a seeded uniform stream instead of thread_rng().

    grid=(-11,11) is the reference's 23x23 lattice (<= 533 spheres);
    grid=(-50,49) is the build-defined "10k" stress scene (BASELINE.json configs[3]).
    Draw order: a', b', selector, then the material's own draws.
    """
    rng = UniformStream(seed)
    world = HittableList()
    world.push(Sphere(Point3(0, -1000, 0), 1000, Lambertian(Color(0.5, 0.5, 0.5))))
    lo, hi = grid
    for a in range(lo, hi + 1):
        for b in range(lo, hi + 1):
            a_prime = float(a) + (0.9 * rng.next())
            b_prime = float(b) + (0.9 * rng.next())
            center = Point3(a_prime, 0.2, b_prime)
            dlt = center - Point3(4, 0.2, 0)
            if math.sqrt(float(dlt[0] * dlt[0] + dlt[1] * dlt[1] + dlt[2] * dlt[2])) > 0.9:
                x = rng.next()
                if 0.0 <= x <= 0.8:
                    c1 = Color(rng.next(), rng.next(), rng.next())
                    c2 = Color(rng.next(), rng.next(), rng.next())
                    mat = Lambertian(c1 * c2)
                elif 0.8 <= x <= 0.95:
                    # random_in_range(0.5, 1): low + (high-low)*u ; gen_range(0.0..0.5): 0.5*u
                    albedo = Color(0.5 + 0.5 * rng.next(), 0.5 + 0.5 * rng.next(), 0.5 + 0.5 * rng.next())
                    fuzz = 0.5 * rng.next()
                    mat = Metal(albedo, fuzz)
                else:
                    mat = Dialectric(1.5)
                world.push(Sphere(center, 0.2, mat))
    world.push(Sphere(Point3(0, 1, 0), 1.0, Dialectric(1.5)))
    world.push(Sphere(Point3(-4, 1, 0), 1.0, Lambertian(Color(0.4, 0.2, 0.1))))
    world.push(Sphere(Point3(4, 1, 0), 1.0, Metal(Color(0.7, 0.6, 0.5), 0.0)))
    return world


def book1_camera(width, height):
    """main.rs:108-118 with ASPECT_RATIO = width/height (a runtime value here)."""
    return Camera(Point3(13, 2, 3), Point3(0, 0, 0), Vec3(0, 1, 0), 20.0,
                  float(width) / float(height), 0.1, 10.0)


# ---- flat scene file: the step immediately before the path ---------------
# Raw little-endian rt_sphere records (72 bytes each, include/rtiow_hip.h), list order kept.
# host/rtiow_render --dump-scene writes the same bytes.

def save_scene(path, world):
    flat = world.flatten() if hasattr(world, "flatten") and not isinstance(world, np.ndarray) else world
    np.ascontiguousarray(flat, dtype=SPHERE_DTYPE).tofile(path)


def load_scene(path):
    import os
    size = os.path.getsize(path)
    if size % SPHERE_DTYPE.itemsize:
        raise ValueError(f"{path}: {size} bytes is not a whole number of 72-byte sphere records")
    flat = np.fromfile(path, dtype=SPHERE_DTYPE)
    if ((flat["kind"] < 0) | (flat["kind"] > 2)).any():
        raise ValueError(f"{path}: unknown material kind")
    return flat
