"""Output stage: the reference's `image.png`, and binary PPM (P6).

The reference saves its frame as an 8-bit RGBA PNG through the `image` crate (main.rs:147,177: `image_buffer.save("image.png")`,
rows top first, the order main.rs:141-145 produces, alpha 255 from `to_rgba`, vec3.rs:403-421).  write_png() writes that file -- the same
pixels in the same order and colour type; the compressed bytes are zlib's, not the crate's encoder's, which no decoder can tell apart --
and write_ppm() the same bytes with alpha dropped as P6.  read_png() decodes what either writer (or the reference) produces: 8-bit
RGB / RGBA, non-interlaced, all five row filters.
"""
import struct
import zlib

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


def _png_chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def write_png(path, rgba_top_first, level=6):
    """RGBA8 PNG (colour type 6, bit depth 8, no interlace, filter 0 on every row): what `image_buffer.save("image.png")` holds."""
    a = np.ascontiguousarray(rgba_top_first, dtype=np.uint8)
    assert a.ndim == 3 and a.shape[2] == 4, "an (H, W, 4) RGBA image"
    h, w = a.shape[0], a.shape[1]
    raw = np.zeros((h, 1 + 4 * w), dtype=np.uint8)               # one filter-type byte (0 = None) before each row
    raw[:, 1:] = a.reshape(h, 4 * w)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(_png_chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)))
        f.write(_png_chunk(b"IDAT", zlib.compress(raw.tobytes(), level)))
        f.write(_png_chunk(b"IEND", b""))


def read_png(path):
    """-> (H, W, C) u8, C = 3 or 4.  8-bit truecolour (with or without alpha), non-interlaced; CRCs checked."""
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n", "not a PNG"
    pos, idat, w = 8, [], None
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] == (zlib.crc32(tag + body) & 0xFFFFFFFF), "bad chunk CRC"
        if tag == b"IHDR":
            w, h, depth, ctype, _, _, interlace = struct.unpack(">IIBBBBB", body)
            assert depth == 8 and ctype in (2, 6) and interlace == 0, "8-bit RGB/RGBA, non-interlaced only"
            c = 3 if ctype == 2 else 4
        elif tag == b"IDAT":
            idat.append(body)
        elif tag == b"IEND":
            break
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8).reshape(h, 1 + c * w)
    out = np.zeros((h, c * w), dtype=np.uint8)
    prev = np.zeros(c * w, dtype=np.int32)
    for y in range(h):
        ft, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:                                           # Up
            cur = (line + prev) & 255
        elif ft == 1:                                           # Sub: a running sum per channel
            cur = (np.cumsum(line.reshape(w, c), axis=0) & 255).reshape(-1)
        else:                                                   # Average / Paeth: sequential in x
            cur = np.zeros(c * w, dtype=np.int32)
            for i in range(c * w):
                a_ = int(cur[i - c]) if i >= c else 0
                b_ = int(prev[i])
                if ft == 3:
                    pred = (a_ + b_) >> 1
                else:
                    c_ = int(prev[i - c]) if i >= c else 0
                    pa, pb, pc = abs(b_ - c_), abs(a_ - c_), abs(a_ + b_ - 2 * c_)
                    pred = a_ if (pa <= pb and pa <= pc) else (b_ if pb <= pc else c_)
                cur[i] = (int(line[i]) + pred) & 255
            assert ft in (3, 4), "unknown row filter"
        out[y] = cur
        prev = cur
    return out.reshape(h, w, c)


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
