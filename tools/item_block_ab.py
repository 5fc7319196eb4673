"""Diagnostic: kernel time of the same launch on work blocks of 256 and of 1 024 pixel-samples (RTIOW_LARGE_BLOCK_MIN_ITEMS moves the
threshold of rt_api.hip: 0 = large blocks whenever the launch has >= 147 (small-grid kernel: 69) samples per pixel, a huge value = never), launches interleaved
in both orders.  usage: python tools/item_block_ab.py"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rtiow_amd as rt
r = rt.Renderer(0)
for name, grid, w, h, spp in [("book 1200x675x500", (-11, 11), 1200, 675, 500), ("book 1200x675x147", (-11, 11), 1200, 675, 147),
                              ("book 1200x675x300", (-11, 11), 1200, 675, 300), ("10k spheres 1920x1080x256", (-50, 49), 1920, 1080, 256)]:
    r.upload_scene(rt.random_scene(1, grid=grid).flatten())
    cam = rt.book1_camera(w, h); p = rt.make_params(w, h, spp)
    d = torch.zeros((h, w, 3), dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    res, crc = {}, {}
    for rnd in range(9):
        for blk, thr in ((("256", str(1 << 62)), ("1024", "0")) if rnd % 2 == 0 else (("1024", "0"), ("256", str(1 << 62)))):
            os.environ["RTIOW_LARGE_BLOCK_MIN_ITEMS"] = thr
            r.render_device(cam, p, d.data_ptr(), stream)
            st = r.last_stats()
            assert bool(st["kernel_variant"] & 4) == (blk == "1024")
            if rnd: res.setdefault(blk, []).append(st["kernel_ms"])
            crc[blk] = int(d.sum().item())
    print(name, {k: round(statistics.median(v), 3) for k, v in res.items()}, "same frame:", crc["256"] == crc["1024"])
os.environ.pop("RTIOW_LARGE_BLOCK_MIN_ITEMS")
