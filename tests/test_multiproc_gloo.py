"""N > 1 plumbing on CPU: world_size-2 gloo processes shard a pair list and gather IBD records to rank 0."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fastsmc_amd import capi
from fastsmc_amd.dist import gather_ibd_records, shard_pair_range


def test_shards_partition_the_pair_list_at_batch_boundaries():
    for n_pairs, world, batch in ((499500, 8, 64), (130, 2, 64), (63, 4, 32), (1, 2, 64)):
        edges = [shard_pair_range(n_pairs, r, world, batch) for r in range(world)]
        assert edges[0][0] == 0 and edges[-1][1] == n_pairs
        for (lo, hi), (lo2, _) in zip(edges, edges[1:]):
            assert hi == lo2 and lo <= hi and (hi % batch == 0 or hi == n_pairs)


def _fake_records(lo, hi, seed):
    """Deterministic per-pair records for local pairs [0, hi-lo): 0..2 segments per pair."""
    rng = np.random.default_rng(seed)
    n_seg = rng.integers(0, 3, size=hi - lo)
    rec = np.zeros(int(n_seg.sum()), capi.IBD_DTYPE)
    rec["pair"] = np.repeat(np.arange(hi - lo, dtype=np.uint32), n_seg)
    rec["start"] = np.arange(rec.size, dtype=np.int32) * 3
    rec["end"] = rec["start"] + 2
    rec["prob"] = rng.random(rec.size, dtype=np.float32)
    return rec


def _worker(rank, world, port, n_pairs, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_pair_range(n_pairs, rank, world, 64)
    rec = _fake_records(lo, hi, seed=100 + rank)
    total, merged = gather_ibd_records(rec, lo, dist, rank, world)
    if rank == 0:
        np.save(out_path, merged)
        assert total == merged.size
    else:
        assert merged is None
    dist.barrier()
    dist.destroy_process_group()


def test_record_gather_world_size_2(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    n_pairs, world = 1000, 2
    out = str(tmp_path / "merged.npy")
    mp.spawn(_worker, args=(world, port, n_pairs, out), nprocs=world, join=True)
    merged = np.load(out)
    want = []
    for r in range(world):
        lo, hi = shard_pair_range(n_pairs, r, world, 64)
        rec = _fake_records(lo, hi, seed=100 + r)
        rec["pair"] += np.uint32(lo)
        want.append(rec)
    want = np.concatenate(want)
    assert merged.dtype == want.dtype and np.array_equal(merged, want)
    assert np.all(np.diff(merged["pair"].astype(np.int64)) >= 0)  # single-device output order


def test_shard_batch_ranges_tile_the_job():
    """HMM.setShard's rule (whole batches, contiguous, nBatches*r/R) restated: shards tile the batch list."""
    from fastsmc_amd.dist import shard_pair_range

    for n_pairs in (0, 1, 31, 32, 33, 2016, 499500):
        for world in (1, 2, 3, 8):
            edges = [shard_pair_range(n_pairs, r, world, batch=32) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n_pairs
            assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
            assert all(lo % 32 == 0 for lo, _ in edges)
