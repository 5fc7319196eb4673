"""Diagnostic: kernel time with work blocks of 256 and of 1 024 pixel-samples (RTIOW_ITEM_BLOCK), launches interleaved in both orders."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rtiow_amd as rt
r = rt.Renderer(0)
for name, grid, w, h, spp in [("book500", (-11, 11), 1200, 675, 500), ("book147", (-11,11), 1200, 675, 147), ("tenk256", (-50, 49), 1920, 1080, 256)]:
    flat = rt.random_scene(1, grid=grid).flatten()
    r.upload_scene(flat)
    cam = rt.book1_camera(w, h); p = rt.make_params(w, h, spp)
    d = torch.zeros((h, w, 3), dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    res = {}
    crc = {}
    for rnd in range(9):
        for blk in (("256", "1024") if rnd % 2 == 0 else ("1024", "256")):
            os.environ["RTIOW_ITEM_BLOCK"] = blk
            r.render_device(cam, p, d.data_ptr(), stream)
            st = r.last_stats()
            if rnd: res.setdefault(blk, []).append(st["kernel_ms"])
            crc[blk] = int(d.sum().item())
    print(name, {k: round(statistics.median(v), 3) for k, v in res.items()}, "same frame:", crc["256"] == crc["1024"])
