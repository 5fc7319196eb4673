"""Generates the committed golden fixtures from the CPU oracles.

    python tests/golden/make_golden.py

  book1_scene_seed1.npy     the flat sphere list of random_scene(seed=1) (72-byte records)
  book1_32x18_4spp.npz      Oracle B exact sums (u64), Oracle A f64 sums, ray counts,
                            RGBA8 bytes for W=32 H=18 spp=4 seed=1 depth=50
  tenk_scene_seed1_head.npy first/last records + count of the 10k stress scene
The oracles are pinned by tests/test_oracle_*.py (Philox KATs, the reference's
PNG sky rows, analytic known answers); these files then guard against drift of
the oracle, the scene builder and the stream addressing.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402
import rtiow_amd as rt  # noqa: E402

flat = rt.random_scene(1).flatten()
np.save(os.path.join(HERE, "book1_scene_seed1.npy"), flat)

W, H, SPP = 32, 18, 4
cam = oracle.book1_camera(W, H)
p = oracle.make_params(W, H, SPP, seed=1)
fix, sm, stb = oracle.render_b(cam, flat, p)
sa, sta = oracle.render_a(cam, flat, p)
np.savez_compressed(os.path.join(HERE, "book1_32x18_4spp.npz"),
                    fix=fix, sum_f32=sm, sum_a_f64=sa,
                    rays_b=np.uint64(stb["rays_traced"]), rays_a=np.uint64(sta["rays_traced"]),
                    depth_hist=np.asarray(stb["depth_hist"], dtype=np.uint64),
                    rgba=oracle.resolve_b(fix, SPP, flip=True))

tenk = rt.random_scene(1, grid=(-50, 49)).flatten()
np.savez_compressed(os.path.join(HERE, "tenk_scene_seed1_head.npz"),
                    count=np.int64(len(tenk)), head=tenk[:8], tail=tenk[-8:],
                    kinds=np.bincount(tenk["kind"], minlength=3))
print("book1 spheres", len(flat), "10k spheres", len(tenk), "rays", stb["rays_traced"])
