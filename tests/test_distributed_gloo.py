"""The N>1 path on CPU: world_size-2 and world_size-8 gloo.  Each rank produces its row shard
(with the CPU oracle standing in for the GPU kernel -- this is a test), one
gather brings the exact sums to rank 0, and the reassembled frame must equal
the single-process frame bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, tile_rows, w, h, spp, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import oracle
    import rtiow_amd as rt
    from rtiow_amd.distributed import FrameGatherer, gather_frame, shard_row_map

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    flat = rt.random_scene(1).flatten()
    cam = oracle.book1_camera(w, h)
    rows = shard_row_map(h, tile_rows, rank, world)
    # render exactly this rank's rows, one oracle call per tile
    parts = []
    for lo in range(0, len(rows), tile_rows):
        tile = rows[lo:lo + tile_rows]
        fix, _, _ = oracle.render_b(cam, flat, oracle.make_params(w, h, spp, rows=(int(tile[0]), int(tile[-1]) + 1, 1), nthreads=2 if world <= 2 else 1))
        parts.append(fix)
    local = np.concatenate(parts, axis=0) if parts else np.zeros((0, w, 3), dtype=np.uint64)
    full = gather_frame(torch.from_numpy(local.view(np.int64).copy()), h, tile_rows, rank, world)
    # the preallocated form bench.py uses, called twice with different data: its buffers are reused
    g = FrameGatherer(h, w, tile_rows, rank, world, "cpu")
    twice = g(torch.from_numpy((local + np.uint64(1)).view(np.int64).copy()))
    again = g(torch.from_numpy(local.view(np.int64).copy()))
    dist.barrier()
    if rank == 0:
        assert twice is again and torch.equal(again, full)
        np.save(out_path, full.numpy().view(np.uint64))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,tile_rows,h", [
    (2, 4, 18), (2, 5, 23),       # 23 rows / tiles of 5: ragged last tile
    (8, 2, 27),                   # the node's 8 ranks: 14 tiles of 2 rows, the last one ragged (1 row) and on rank 5; ranks 6, 7 own ONE tile,
                                  # the others two: unequal shards behind equally padded send buffers
    (8, 1, 11),                   # tiles of one row (bench.py's default): ranks 3..7 own one row, ranks 0..2 two
])
def test_shard_gather_equals_single_process(tmp_path, oracle_mod, book1_flat, world, tile_rows, h):
    import torch.multiprocessing as mp

    w, spp = (32, 3) if world == 2 else (16, 2)
    out = str(tmp_path / "full.npy")
    port = _free_port()
    mp.spawn(_worker, args=(world, port, tile_rows, w, h, spp, out), nprocs=world, join=True)
    got = np.load(out)
    want, _, _ = oracle_mod.render_b(oracle_mod.book1_camera(w, h), book1_flat, oracle_mod.make_params(w, h, spp))
    assert np.array_equal(got, want)
