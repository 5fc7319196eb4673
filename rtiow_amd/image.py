"""Output stage: binary PPM (P6) writer.

The reference saves RGBA8 PNG through the `image` crate (main.rs:147,177); the
build writes the same bytes (alpha dropped) as P6, top row first, which is the
order main.rs:141-145 produces.
"""
import numpy as np


def write_ppm(path, rgba_top_first):
    a = np.asarray(rgba_top_first, dtype=np.uint8)
    h, w = a.shape[0], a.shape[1]
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (w, h))
        f.write(np.ascontiguousarray(a[:, :, :3]).tobytes())


def read_ppm(path):
    with open(path, "rb") as f:
        data = f.read()
    parts = data.split(b"\n", 3)
    assert parts[0] == b"P6"
    w, h = (int(x) for x in parts[1].split())
    return np.frombuffer(parts[3], dtype=np.uint8).reshape(h, w, 3)


# ---- resumable accumulation -----------------------------------------------
# The exact u64 sums are additive over sample ranges (counter-based RNG + integer sums), so a
# checkpoint is just the sums plus how many samples they hold; resuming = rendering the next
# sample range (rt_params.sample_begin) and adding.

def save_checkpoint(path, fix, spp_done, seed):
    np.savez_compressed(path, fix=np.ascontiguousarray(fix, dtype=np.uint64),
                        spp_done=np.int64(spp_done), seed=np.uint64(seed))


def load_checkpoint(path):
    z = np.load(path, allow_pickle=False)
    return z["fix"], int(z["spp_done"]), int(z["seed"])
