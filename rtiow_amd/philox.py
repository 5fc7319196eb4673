"""Philox4x32-10 on the host (pure Python integers).

Used only to SEED THE SCENE BUILDER reproducibly: the reference builds its
scene from rand::thread_rng() (src/main.rs:60), which cannot be replayed.
The render path's own Philox lives in the HIP kernel (csrc/rt_device.hpp).
"""
M0, M1 = 0xD2511F53, 0xCD9E8D57
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF


def philox4x32_10(ctr, key):
    c0, c1, c2, c3 = ctr
    k0, k1 = key
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & MASK, p1 & MASK, ((p0 >> 32) ^ c3 ^ k1) & MASK, p0 & MASK
        k0 = (k0 + W0) & MASK
        k1 = (k1 + W1) & MASK
    return c0, c1, c2, c3


class UniformStream:
    """Sequential U[0,1) draws, u = (word >> 8) * 2^-24, from counter (block, 0, 0, tag)."""

    def __init__(self, seed, tag=0x5CE9E):
        self.key = (seed & MASK, (seed >> 32) & MASK)
        self.tag = tag
        self.block = 0
        self.words = ()
        self.pos = 0

    def next(self):
        if self.pos >= len(self.words):
            self.words = philox4x32_10((self.block & MASK, 0, 0, self.tag), self.key)
            self.block += 1
            self.pos = 0
        w = self.words[self.pos]
        self.pos += 1
        return (w >> 8) * (1.0 / 16777216.0)
