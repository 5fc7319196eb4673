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


def gather_frame(local_fix, height, tile_rows, rank, world, dst=0, group=None):
    """local_fix: torch int64 tensor [rows_k, W, 3] holding this rank's exact sums
    (the u64 bit patterns).  Returns the full [H, W, 3] frame on rank `dst`, None
    elsewhere.  One collective: torch.distributed.gather of equally padded tiles."""
    import torch
    import torch.distributed as dist

    width = local_fix.shape[1]
    pad_rows = max_shard_rows(height, tile_rows, world)
    padded = torch.zeros((pad_rows, width, 3), dtype=torch.int64, device=local_fix.device)
    padded[: local_fix.shape[0]] = local_fix
    if world == 1:
        parts = [padded]
    else:
        parts = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
        dist.gather(padded, gather_list=parts, dst=dst, group=group)
    if rank != dst:
        return None
    full = torch.empty((height, width, 3), dtype=torch.int64, device=local_fix.device)
    for k in range(world):
        rows = shard_row_map(height, tile_rows, k, world)
        idx = torch.as_tensor(rows, device=local_fix.device)
        full.index_copy_(0, idx, parts[k][: len(rows)])
    return full
