"""Progressive passes against one launch: 1200x675, 500 samples per pixel as ONE launch, as 5 passes of 100 on
one stream (each pays its end-of-launch tail), and as 5 passes of 100 alternating between two streams (pass k + 1 fills pass k's
tail: the context holds two launches' state).  Wall time between synchronisations, best and median of N frames; the three frames
must be bit-identical (exact integer sums).   usage: python tools/progressive_bench.py [frames]"""
import os, statistics, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtiow_amd as rt

N = int(sys.argv[1]) if len(sys.argv) > 1 else 7
OVERLAPPED_FLAG = os.environ.get("NO_OVERLAPPED_FLAG") is None      # (NO_OVERLAPPED_FLAG=1: the two-stream passes without RT_FLAG_OVERLAPPED, i.e. on blocks of 256)
W, H = 1200, 675
r = rt.Renderer(0)
r.upload_scene(rt.random_scene(1).flatten())
cam = rt.book1_camera(W, H)
s = [torch.cuda.Stream(), torch.cuda.Stream()]
d_fix = torch.zeros((H, W, 3), dtype=torch.int64, device="cuda")


def frame(passes, spp_pass, two_streams, spp_list=None):
    d_fix.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    b = 0
    for k in range(passes):
        n = spp_list[k] if spp_list else spp_pass
        p = rt.make_params(W, H, n, sample_begin=b, seed=1, flags=rt.RT_FLAG_ACCUMULATE | (rt.RT_FLAG_OVERLAPPED if two_streams and OVERLAPPED_FLAG else 0))
        r.render_device(cam, p, d_fix.data_ptr(), s[k & 1 if two_streams else 0].cuda_stream)
        b += n
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


rows = []
crcs = set()
for name, args in (("one launch of 500 spp", (1, 500, False)), ("5 passes of 100 spp, one stream", (5, 100, False)),
                   ("5 passes of 100 spp, two streams alternating", (5, 100, True)),
                   ("10 passes of 50 spp, two streams alternating", (10, 50, True)),
                   ("2 passes of 250 spp, two streams", (2, 250, True))):
    frame(*args)
    ts = [frame(*args) for _ in range(N)]
    crcs.add(zlib.crc32(d_fix.cpu().numpy().tobytes()))
    rows.append((name, min(ts), statistics.median(ts)))
base = rows[0][2]
for name, best, med in rows:
    print(f"{name:48s} median {med:8.3f} ms  best {best:8.3f} ms  = {med / base:6.4f} x the single launch   {W * H * 500 / med / 1e3:7.1f} Msamples/s")
print("frames bit-identical:", len(crcs) == 1)
