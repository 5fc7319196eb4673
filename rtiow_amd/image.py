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
