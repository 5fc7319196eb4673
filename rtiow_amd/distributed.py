"""Row sharding across GPUs and the single framebuffer gather.

The reference parallelises over image rows with rayon (main.rs:122-123) and
`collect()` concatenates the rows in order (main.rs:139).  Here the rows are
cut into tiles of `tile_rows` rows dealt round-robin to the ranks (sky rows
cost ~1 ray per sample, ground rows ~3: contiguous blocks would unbalance the
ranks), every rank renders its tiles with no data-path communication, and ONE
gather (RCCL over xGMI when the tensors are on GPUs, gloo on CPU) brings the
exact sums to rank 0, which puts the rows back in image order.

Because the Philox counter is keyed by the GLOBAL pixel index and the sample
index, and pixel sums are exact integers, the assembled frame is bit-identical
to a single-GPU render of the whole image.
"""
import numpy as np


def shard_row_map(height, tile_rows, shard_index, shard_count):
    """Image rows j (0 = bottom) owned by a shard, ascending: the compact-row order
    of rt_render's output (include/rtiow_hip.h, rt_shard_row_index)."""
    ntiles = (height + tile_rows - 1) // tile_rows
    rows = []
    for t in range(shard_index, ntiles, shard_count):
        lo = t * tile_rows
        rows.extend(range(lo, min(lo + tile_rows, height)))
    return np.asarray(rows, dtype=np.int64)


def max_shard_rows(height, tile_rows, shard_count):
    return max(len(shard_row_map(height, tile_rows, k, shard_count)) for k in range(shard_count))


class FrameGatherer:
    """The gather of one frame geometry, with every buffer and row-index tensor allocated once:
    per frame it costs one copy into the padded send buffer, ONE torch.distributed.gather, and one
    index_copy per rank on `dst` (no host-side work in the loop other than launching those)."""

    def __init__(self, height, width, tile_rows, rank, world, device, dst=0, group=None):
        import torch
        self.height, self.width, self.tile_rows = int(height), int(width), int(tile_rows)
        self.rank, self.world, self.dst, self.group = int(rank), int(world), int(dst), group
        self.rows_local = len(shard_row_map(height, tile_rows, rank, world))
        pad_rows = max_shard_rows(height, tile_rows, world)
        self.padded = torch.zeros((pad_rows, width, 3), dtype=torch.int64, device=device)
        self.parts = self.full = self.idx = None
        if rank == dst:
            self.parts = [self.padded] if world == 1 else [torch.empty_like(self.padded) for _ in range(world)]
            self.full = torch.empty((height, width, 3), dtype=torch.int64, device=device)
            self.idx = [torch.as_tensor(shard_row_map(height, tile_rows, k, world), device=device) for k in range(world)]

    def __call__(self, local_fix):
        """local_fix: int64 [rows_k, W, 3] (the u64 bit patterns of this rank's exact sums).
        Returns the full [H, W, 3] frame on rank `dst` (a buffer reused by the next call), None elsewhere."""
        import torch.distributed as dist
        assert local_fix.shape[0] == self.rows_local and local_fix.shape[1] == self.width
        self.padded[: self.rows_local].copy_(local_fix)
        if self.world > 1:
            dist.gather(self.padded, gather_list=self.parts, dst=self.dst, group=self.group)
        if self.rank != self.dst:
            return None
        for k in range(self.world):
            self.full.index_copy_(0, self.idx[k], self.parts[k][: self.idx[k].shape[0]])
        return self.full


def gather_frame(local_fix, height, tile_rows, rank, world, dst=0, group=None):
    """local_fix: torch int64 tensor [rows_k, W, 3] holding this rank's exact sums
    (the u64 bit patterns).  Returns the full [H, W, 3] frame on rank `dst`, None
    elsewhere.  One collective: torch.distributed.gather of equally padded tiles.
    (One-shot form of FrameGatherer.)"""
    g = FrameGatherer(height, local_fix.shape[1], tile_rows, rank, world, local_fix.device, dst=dst, group=group)
    return g(local_fix)
