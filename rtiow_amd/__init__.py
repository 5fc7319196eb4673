"""rtiow_amd -- MI355X-native hot path of Druthyn/rtiow.

The package holds only what the path needs: the HIP megakernel and its C ABI
(csrc/, built into librtiow_hip.so), a ctypes binding, and a host-side mirror
of the reference's scene interface (Camera, Sphere, Lambertian, Metal,
Dialectric, HittableList, random_scene).  See DESIGN.md.
"""
from .scene import (Camera, Color, Dialectric, HittableList, Lambertian, Metal, Point3, Scatter,
                    Sphere, Vec3, book1_camera, random_scene, save_scene, load_scene, SPHERE_DTYPE)
from .render import Renderer, make_params, shard_rows, shard_row_indices, tube_tile_host, tile_layout_host
from .image import write_ppm, read_ppm, write_png, read_png, save_checkpoint, load_checkpoint
from ._ffi import RtiowHipError, RT_FLAG_ACCUMULATE, RT_FLAG_NO_FILTER, RT_FLAG_DIAG_STATS, RT_FLAG_UNIFORM53, RT_FLAG_OVERLAPPED

__all__ = ["Camera", "Color", "Dialectric", "HittableList", "Lambertian", "Metal", "Point3", "Scatter",
           "Sphere", "Vec3", "book1_camera", "random_scene", "SPHERE_DTYPE", "Renderer", "make_params",
           "shard_rows", "shard_row_indices", "tube_tile_host", "tile_layout_host", "write_ppm", "read_ppm", "write_png", "read_png", "save_scene", "load_scene", "save_checkpoint", "load_checkpoint", "RtiowHipError", "RT_FLAG_ACCUMULATE", "RT_FLAG_NO_FILTER", "RT_FLAG_DIAG_STATS", "RT_FLAG_UNIFORM53", "RT_FLAG_OVERLAPPED"]
