"""N > 1 plumbing on CPU: world_size-2 gloo processes shard a pair list and gather IBD records to rank 0."""
import os
import socket

import numpy as np
import pytest
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


# ---------------------------------------------------------------------------------------------------------------
# bench.py --gpus N (strong scaling of one sharded work list): the exact sharding + gather code path, world 2, gloo

def test_all_pairs_at_matches_the_enumeration():
    import bench
    from fastsmc_amd.dist import all_pairs_at, sample_pair_ordinals

    full = bench.all_pairs(37)  # HMM::decodeAll order (HMM.cpp:325-357)
    np.testing.assert_array_equal(all_pairs_at(np.arange(full.shape[0])), full)
    o = sample_pair_ordinals(37, 500, seed=3)
    assert o.size == 500 and np.all(np.diff(o) > 0) and o[-1] < full.shape[0]
    np.testing.assert_array_equal(sample_pair_ordinals(37, 500, seed=3), o)  # seeded: every rank draws the same list
    assert sample_pair_ordinals(5, 10 ** 6, seed=1).size == 45  # more than exist: the whole enumeration


def test_weight_shards_tile_the_groups_and_balance():
    from fastsmc_amd.dist import shard_groups_by_weight

    rng = np.random.default_rng(0)
    for n_groups, world in ((1, 2), (7, 2), (1000, 8), (16384, 8), (5, 8)):
        for w in (np.ones(n_groups), rng.integers(320, 5504, size=n_groups) * 32.0):  # whole-sequence / hashing windows
            edges = [shard_groups_by_weight(w, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n_groups
            assert all(a[1] == b[0] and a[0] <= a[1] for a, b in zip(edges, edges[1:]))
            if n_groups >= 100 * world:
                tot = [w[lo:hi].sum() for lo, hi in edges]
                assert max(tot) / (sum(tot) / world) < 1.02


def _strong_worker(rank, world, port, n_ind, n_list, S, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fastsmc_amd.dist import all_pairs_at, sample_pair_ordinals, shard_groups_by_weight

    ordinals = sample_pair_ordinals(n_ind, n_list, 20260)
    groups = capi.whole_sequence_groups(ordinals.size, S, batch=64)
    weights = groups["n_pairs"].astype(np.float64) * (groups["to"] - groups["from"])
    g_lo, g_hi = shard_groups_by_weight(weights, rank, world)
    lo = int(groups["first_pair"][g_lo]) if g_lo < groups.size else ordinals.size
    hi = int(groups["first_pair"][g_hi]) if g_hi < groups.size else ordinals.size
    pairs = all_pairs_at(ordinals[lo:hi])
    assert pairs.shape[0] == hi - lo and (lo % 64 == 0)
    rec = _fake_records(lo, hi, seed=7 + lo)  # stands in for the decode of this shard (local pair indices)
    total, merged = gather_ibd_records(rec, lo, dist, rank, world)
    if rank == 0:
        np.save(out_path, merged)
        np.save(out_path + ".edges.npy", np.array([lo, hi]))
    dist.barrier()
    dist.destroy_process_group()


def test_strong_scaling_list_world_size_2(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    n_ind, n_list, S, world = 30, 1000, 640, 2
    out = str(tmp_path / "merged.npy")
    mp.spawn(_strong_worker, args=(world, port, n_ind, n_list, S, out), nprocs=world, join=True)
    merged = np.load(out)
    lo0, hi0 = np.load(out + ".edges.npy")
    assert lo0 == 0 and hi0 % 64 == 0 and abs(hi0 - n_list / 2) <= 64  # equal weight: half the list, whole groups
    want = []
    for lo, hi in ((0, int(hi0)), (int(hi0), n_list)):
        rec = _fake_records(lo, hi, seed=7 + lo)
        rec["pair"] += np.uint32(lo)
        want.append(rec)
    want = np.concatenate(want)
    assert np.array_equal(merged, want) and np.all(np.diff(merged["pair"].astype(np.int64)) >= 0)


# the product's in-memory gather (fastsmc_amd.dist.gather_hmm_records): the record columns an HMM keeps, world 2, gloo
class _FakeHmm:
    def __init__(self, rec):
        self._rec = rec

    def getIbdRecordArrays(self):
        return {n: self._rec[n] for n in self._rec.dtype.names}


def _fake_rows(n, seed):
    from fastsmc_amd.dist import IBD_ROW_DTYPE

    rng = np.random.default_rng(seed)
    rec = np.zeros(n, IBD_ROW_DTYPE)
    rec["pair"] = np.sort(rng.integers(0, 500, size=n))
    rec["hap_a"], rec["hap_b"] = rng.integers(0, 64, size=n), rng.integers(64, 128, size=n)
    rec["start"] = rng.integers(0, 1000, size=n)
    rec["end"] = rec["start"] + rng.integers(0, 50, size=n)
    rec["prob"], rec["post_mean"], rec["map"] = rng.random(n), rng.random(n) * 100, rng.random(n) * 100
    return rec


def _hmm_worker(rank, world, port, out_path):
    from fastsmc_amd.dist import gather_hmm_records

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total, merged = gather_hmm_records(_FakeHmm(_fake_rows(37 if rank == 0 else 0 if rank == 1 else 11, 7 + rank)), dist,
                                       rank, world)
    if rank == 0:
        np.save(out_path, merged)
        assert total == merged.size
    else:
        assert merged is None
    dist.barrier()
    dist.destroy_process_group()


def test_hmm_record_gather_world_size_3_with_an_empty_rank(tmp_path):
    from fastsmc_amd.dist import gather_hmm_records

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "rows.npy")
    mp.spawn(_hmm_worker, args=(3, port, out), nprocs=3, join=True)
    merged = np.load(out)
    want = np.concatenate([_fake_rows(37, 7), _fake_rows(0, 8), _fake_rows(11, 9)])
    assert merged.dtype == want.dtype and np.array_equal(merged, want)
    # one process: the records as they are
    total, alone = gather_hmm_records(_FakeHmm(_fake_rows(5, 1)))
    assert total == 5 and np.array_equal(alone, _fake_rows(5, 1))


def test_bench_start_up_of_two_ranks_builds_the_cohort_once(tmp_path):
    """bench.py at N = 2 (the driver's launch line, gloo instead of RCCL, `--startup-only`: everything up to the first
    GPU call): local rank 0 synthesises and prepares the cohort and leaves it in the node's cache, rank 1 maps it; the
    host threads of a rank are capped at cores / ranks; both ranks end up with their shard of the one work list."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, FSMC_BENCH_CACHE=str(tmp_path / "cache"))
    env.pop("FSMC_HOST_THREADS", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo",
           "--workload", "c3", "--haps", "64", "--sites", "700", "--pairs", "1500", "--startup-only"]
    lines = []
    for _ in range(2):  # the second run finds the cache file
        r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=root, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        lines.append(json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]))
    files = os.listdir(tmp_path / "cache")
    assert len(files) == 1 and files[0].startswith("cohort_") and files[0].endswith(".npz")
    for line in lines:
        assert line["startup_only"] and line["n_ranks"] == 2 and len(line["startup_s_per_rank"]) == 2
        assert 0 < line["pairs_of_rank_0"] < 1500  # its shard of the one list
        cores = len(os.sched_getaffinity(0))
        assert line["host_threads_per_rank"] == str(max(1, cores // 2))


# ---------------------------------------------------------------------------------------------------------------
# sum-over-pairs mode across ranks (fastsmc_amd.dist.reduce_sums): the reference's jobs + PosteriorMerger, world 2 / 3

def _sums_problem():
    from fastsmc_amd import synth
    from oracle import oracle as O

    tables = synth.make_model_tables(12)
    haps = synth.make_haps(64, 48, seed=13, cm_per_mb=25.0, switch_per_cm=0.6)  # (16 of its 32 individuals are decoded)
    _, derived, flipped = synth.fold_and_pack(haps.alleles)
    folded = np.where(flipped[None, :], 1 - haps.alleles, haps.alleles).astype(np.uint8)
    pm = O.prepare_model(tables, (haps.cm / 100.0).astype(np.float32), haps.bp, derived, 64, time=50)
    return pm, folded


def _job_sums(pm, folded, jobs, job_ind, batch=64):
    """The four posterior-sum planes HMM::decodeAll(jobs, jobInd) leaves (HMM.cpp:310-357, 1044-1085), by the oracle:
    the job's pairs in enumeration order, batches of `batch`, every batch's local sum added to the planes in turn."""
    from oracle import oracle as O

    pairs = O.enumerate_all_pairs(16, jobs, job_ind)
    planes = [np.zeros((pm.S, pm.K), np.float32) for _ in range(4)]
    for lo in range(0, len(pairs), batch):
        sub = pairs[lo:lo + batch]
        ob = np.stack([folded[a] ^ folded[b] for a, b in sub])
        hb = np.stack([folded[a] & folded[b] for a, b in sub])
        post, _ = O.decode_batch(pm, ob, hb, 0, pm.S)
        O.augment_sum_over_pairs(pm, post, len(sub), ob, hb, *planes)
    return planes


def _sums_worker(rank, world, port, out_path, order):
    from fastsmc_amd.dist import SUM_PLANES, reduce_sums

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pm, folded = _sums_problem()
    mine = _job_sums(pm, folded, world, rank + 1)
    total = reduce_sums(dict(zip(SUM_PLANES, mine)), dist, rank, world, order=order)
    if order == "rank":
        assert (total is None) == (rank != 0)
    if total is not None:
        np.savez(f"{out_path}.{rank}.npz", **total)
    np.savez(f"{out_path}.part{rank}.npz", *mine)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sum_over_pairs_reduced_in_rank_order(tmp_path, world):
    """Every rank decodes job rank+1 of `world` (the reference's decomposition) and `reduce_sums` merges the planes on
    rank 0 job after job, PosteriorMerger's order: EXACTLY ((0 + P_0) + P_1) + ..., and within 1e-6 relative of the sums
    of one process that decodes all the pairs (the only difference is fp32 re-association at the shard joins)."""
    from fastsmc_amd.dist import SUM_PLANES

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "sums")
    mp.spawn(_sums_worker, args=(world, port, out, "rank"), nprocs=world, join=True)
    got = np.load(f"{out}.0.npz")
    assert not os.path.exists(f"{out}.1.npz")
    parts = [[np.load(f"{out}.part{r}.npz")[f"arr_{i}"] for i in range(4)] for r in range(world)]
    pm, folded = _sums_problem()
    single = _job_sums(pm, folded, 1, 1)
    for i, name in enumerate(SUM_PLANES):
        acc = np.zeros_like(parts[0][i])
        for r in range(world):
            acc = acc + parts[r][i]
        np.testing.assert_array_equal(got[name], acc)             # the merge order, bit for bit
        np.testing.assert_allclose(got[name], single[i], rtol=1e-6, atol=1e-7)  # one process: re-association only
        assert single[i].any()
    np.testing.assert_allclose(got["sumOverPairs"], got["sumOverPairs00"] + got["sumOverPairs01"] + got["sumOverPairs11"],
                               rtol=1e-5, atol=1e-6)


def test_sum_over_pairs_all_reduce_variant(tmp_path):
    """`order="allreduce"`: every rank gets the total; flagged as re-associating -- compared within tolerance only."""
    from fastsmc_amd.dist import SUM_PLANES, reduce_sums

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "sums")
    mp.spawn(_sums_worker, args=(2, port, out, "allreduce"), nprocs=2, join=True)
    a, b = np.load(f"{out}.0.npz"), np.load(f"{out}.1.npz")
    pm, folded = _sums_problem()
    single = _job_sums(pm, folded, 1, 1)
    for i, name in enumerate(SUM_PLANES):
        np.testing.assert_array_equal(a[name], b[name])
        np.testing.assert_allclose(a[name], single[i], rtol=1e-6, atol=1e-7)
    # one process, no group: the planes as they are; planes a decode did not ask for are left out
    alone = reduce_sums({"sumOverPairs": single[0], "sumOverPairs00": np.zeros((0, 0), np.float32)})
    assert list(alone) == ["sumOverPairs"] and np.array_equal(alone["sumOverPairs"], single[0])
    with pytest.raises(ValueError):
        reduce_sums(single, order="tree")
