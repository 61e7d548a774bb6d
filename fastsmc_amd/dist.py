"""Multi-GPU plumbing for the decode path: one process per GPU (``torch.distributed``; backend "nccl" is
RCCL on ROCm, "gloo" in CPU tests).  Pairs are independent, so the pair list is sharded with no collective
on the data path; the only exchanges are the final gather of variable-length IBD records to rank 0 (FastSMC mode)
and the reduction of the per-rank posterior sums (ASMC mode, ``reduce_sums``)."""
from __future__ import annotations

import numpy as np


def shard_pair_range(n_pairs: int, rank: int, world: int, batch: int = 64) -> tuple[int, int]:
    """Contiguous shard [lo, hi) of the pair list for this rank, cut at batch boundaries so that every
    reference batch (HMM.cpp:555-591) stays on one device.  Same split rule as the reference's job ranges
    (pairs * r / R, HMM.cpp:319-321), applied to batches."""
    n_batches = (n_pairs + batch - 1) // batch
    lo_b = n_batches * rank // world
    hi_b = n_batches * (rank + 1) // world
    return min(lo_b * batch, n_pairs), min(hi_b * batch, n_pairs)


def shard_groups_by_weight(weights, rank: int, world: int) -> tuple[int, int]:
    """Contiguous shard [lo, hi) of a list of work-list groups such that every rank gets (as nearly as whole groups
    allow) the same total weight -- the weight of a group is its pair-sites, pairs x decode-window length, so that
    hashing-mode lists with windows of 320...5504 sites balance as well as whole-sequence lists.
    Shard r ends at the first group boundary whose prefix weight reaches total * (r + 1) / world (the reference's
    own range rule, HMM.cpp:319-321, applied to weight instead of pair count)."""
    w = np.asarray(weights, dtype=np.float64)
    if w.size == 0:
        return 0, 0
    prefix = np.concatenate([[0.0], np.cumsum(w)])
    total = prefix[-1]

    def cut(r: int) -> int:
        if r <= 0:
            return 0
        if r >= world:
            return int(w.size)
        return int(np.searchsorted(prefix, total * r / world, side="left"))

    return cut(rank), cut(rank + 1)


def all_pairs_at(index: np.ndarray) -> np.ndarray:
    """Rows (hapA, hapB) of the pairs with the given ordinals in HMM::decodeAll's enumeration (HMM.cpp:325-357):
    individual i contributes the 4*i cross pairs with every j < i -- (jHap, iHap) = (1,1), (2,1), (1,2), (2,2), the
    lower-numbered individual first -- and then its own two haplotypes, so its block starts at ordinal 2*i*i - i."""
    x = np.asarray(index, dtype=np.int64)
    i = ((1.0 + np.sqrt(1.0 + 8.0 * x.astype(np.float64))) / 4.0).astype(np.int64)
    i = np.where(2 * i * i - i > x, i - 1, i)  # guard the float square root at block boundaries
    i = np.where(2 * (i + 1) * (i + 1) - (i + 1) <= x, i + 1, i)
    r = x - (2 * i * i - i)
    within = r == 4 * i
    j = np.where(within, i, r // 4)
    c = r % 4
    hap_a = np.where(within, 2 * i, 2 * j + (c % 2))
    hap_b = np.where(within, 2 * i + 1, 2 * i + (c // 2))
    return np.stack([hap_a, hap_b], axis=1).astype(np.uint32)


def sample_pair_ordinals(n_individuals: int, n_pairs: int, seed: int) -> np.ndarray:
    """A fixed, seeded sub-list of the all-pairs enumeration, in enumeration order (sorted ordinals)."""
    total = 2 * n_individuals * n_individuals - n_individuals
    if n_pairs >= total:
        return np.arange(total, dtype=np.int64)
    rng = np.random.default_rng(seed)
    # distinct ordinals without materialising the whole range: draw with a margin, keep the first n_pairs distinct
    got = np.unique(rng.integers(0, total, size=int(n_pairs * 1.05) + 1024, dtype=np.int64))
    while got.size < n_pairs:
        got = np.unique(np.concatenate([got, rng.integers(0, total, size=n_pairs, dtype=np.int64)]))
    keep = np.sort(rng.permutation(got.size)[:n_pairs])
    return got[keep]


def _gather_records(local: np.ndarray, dist, rank: int, world: int, device):
    """The path's one exchange: ``all_gather`` of the per-rank record counts, then ``gather`` of the byte payloads
    padded to the longest one, to rank 0 (~34 B a record: latency-bound, topology irrelevant).  Returns
    (total_count, concatenation in rank order on rank 0 / None elsewhere)."""
    import torch

    cnt = torch.tensor([local.size], device=device, dtype=torch.int64)
    counts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt)
    counts = [int(c.item()) for c in counts]
    item = local.dtype.itemsize
    payload = torch.zeros(max(max(counts), 1) * item, dtype=torch.uint8, device=device)
    if local.size:
        payload[: local.nbytes] = torch.from_numpy(local.view(np.uint8).reshape(-1).copy()).to(device)
    bucket = [torch.empty_like(payload) for _ in range(world)] if rank == 0 else None
    dist.gather(payload, bucket, dst=0)
    if rank != 0:
        return sum(counts), None
    parts = [bucket[r][: counts[r] * item].cpu().numpy().view(local.dtype) for r in range(world)]
    return sum(counts), np.concatenate(parts)


def gather_ibd_records(rec: np.ndarray, pair_offset: int, dist, rank: int, world: int, device="cpu",
                       force_collective: bool = False):
    """Gather every rank's IBD records (structured array with a ``pair`` field holding *local* pair indices) to
    rank 0: all_gather of counts, then gather of padded byte payloads.  Returns (total_count, records_or_None);
    on rank 0 the records carry global pair indices and are ordered like a single-device run (shards are contiguous
    and each is already ordered, so the concatenation in rank order IS that order).  A group of one rank returns its
    own records without a collective unless ``force_collective`` (the one-rank rehearsal of the RCCL leg)."""
    local = rec.copy()
    local["pair"] += np.uint32(pair_offset)
    if dist is None or (world == 1 and not force_collective):
        return int(local.size), local
    return _gather_records(local, dist, rank, world, device)


def run_fastsmc_sharded(params, rank: int | None = None, world: int | None = None, local_rank: int | None = None,
                        barrier=None, gather: str = "auto", force_collective: bool = False) -> str | None:
    """FastSMC.run() of one job spread over ``world`` GPUs, one process each (launch with
    ``python -m torch.distributed.run --nproc-per-node N ...`` or pass rank/world explicitly).  Every rank loads
    the same inputs and decodes a contiguous range of the job's batches on device ``local_rank``; no collective on the
    data path.  How the records reach the one output file:

    * ``gather="records"`` -- the path's one exchange over the process group: every rank keeps its records in memory
      (no part file), ``gather_hmm_records`` sends them to rank 0 -- RCCL over xGMI when the group's backend is "nccl"
      (payloads staged on ``cuda:local_rank``), gloo otherwise -- and rank 0 writes the file through the product's own
      formatter (``HMM.writeIbdRecordArrays``): the bytes a single-GPU run decompresses to.
    * ``gather="files"`` -- every rank writes ``<output>.part<rank>of<world>``; after a barrier rank 0 concatenates the
      parts in rank order (gzip members concatenate into a valid stream whose content is byte-identical to the
      single-GPU output) and removes them.  Needs a file system the ranks share; the fallback when there is no
      process group to send records through (an explicit ``barrier`` callable).
    * ``gather="auto"`` (default): "records" when ``torch.distributed`` is initialised (or is brought up here), "files"
      when the caller passed its own ``barrier``.

    An initialised process group is used as it is (backend "nccl" = RCCL: bring it up BEFORE the first other GPU call
    of the process, tests/test_gpu_rccl_one_rank.py); without one, and without ``barrier``, a gloo group is brought up
    here.  ``force_collective``: a world of ONE rank takes the "records" route all the same, through the initialised
    group's collectives (the one-rank rehearsal of the RCCL leg on a one-GPU box).  Returns the output file name on
    rank 0, None elsewhere."""
    import os
    import shutil

    from . import api

    if gather not in ("auto", "records", "files"):
        raise ValueError("gather is 'auto', 'records' or 'files'")
    rank = int(os.environ.get("RANK", 0)) if rank is None else rank
    world = int(os.environ.get("WORLD_SIZE", 1)) if world is None else world
    local_rank = int(os.environ.get("LOCAL_RANK", rank)) if local_rank is None else local_rank
    dist = None
    if (world > 1 or force_collective) and (barrier is None or gather == "records"):
        import torch.distributed as dist

        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("gloo", rank=rank, world_size=world)
        if barrier is None:
            barrier = dist.barrier
    if gather == "auto":
        gather = "records" if dist is not None else "files"
    params.gpuDevice = local_rank
    f = api.FastSMC(params)
    f.setShard(rank, world)
    part = f.outputFileName()
    if world == 1 and not force_collective:
        f.run()
        return part
    final = part[: part.rindex(".part")] if ".part" in part else part
    if gather == "records":
        device = "cpu"
        if dist.get_backend() == "nccl":
            import torch

            torch.cuda.set_device(local_rank)
            device = "cuda"
        hmm = f.hmm()
        hmm.setKeepIbdRecords(True)
        hmm.setWriteIbdFile(False)
        f.run()
        _, rows = gather_hmm_records(hmm, dist, rank, world, device=device, force_collective=force_collective)
        if rank == 0:
            hmm.writeIbdRecordArrays(final, rows["hap_a"], rows["hap_b"], rows["start"], rows["end"], rows["prob"],
                                     rows["post_mean"], rows["map"])
        barrier()
        return final if rank == 0 else None
    f.run()
    barrier()
    if rank == 0:
        with open(final, "wb") as out:
            for r in range(world):
                name = f"{final}.part{r}of{world}"
                with open(name, "rb") as src:
                    shutil.copyfileobj(src, out)
                os.remove(name)
    barrier()
    return final if rank == 0 else None


IBD_COLUMNS = (("pair", np.uint64), ("hap_a", np.uint32), ("hap_b", np.uint32), ("start", np.int32), ("end", np.int32),
               ("prob", np.float32), ("post_mean", np.float32), ("map", np.float32))
IBD_ROW_DTYPE = np.dtype([(n, np.dtype(t).newbyteorder("<")) for n, t in IBD_COLUMNS])


def gather_hmm_records(hmm, dist=None, rank: int = 0, world: int = 1, device="cpu", force_collective: bool = False):
    """The in-memory counterpart of ``run_fastsmc_sharded``'s part files: every rank's kept IBD records
    (``HMM.setKeepIbdRecords(True)``; ``HMM.getIbdRecordArrays()``) gathered to rank 0 over the process group -- RCCL over
    xGMI with backend "nccl" and ``device="cuda"``, gloo in CPU tests -- as one structured array (``IBD_ROW_DTYPE``).
    Shards are contiguous ranges of the job's batches and each rank's records are in output order, so the
    concatenation in rank order IS the single-GPU record stream (``pair`` = the record's pair ordinal within its rank's
    shard).  Returns (total_count, records on rank 0 / None elsewhere).  The only collective of the path: an all_gather
    of the counts and one gather of padded payloads (``_gather_records``); a group of one rank skips it unless
    ``force_collective``."""
    cols = hmm.getIbdRecordArrays()
    local = np.zeros(cols["pair"].size, IBD_ROW_DTYPE)
    for name, _ in IBD_COLUMNS:
        local[name] = cols[name]
    if dist is None or (world == 1 and not force_collective):
        return int(local.size), local
    return _gather_records(local, dist, rank, world, device)


SUM_PLANES = ("sumOverPairs", "sumOverPairs00", "sumOverPairs01", "sumOverPairs11")


def _sum_planes(source) -> dict:
    """The posterior-sum planes a rank holds, as float32 [sites][states] arrays: an HMM (its DecodingReturnValues), a
    DecodingReturnValues, or a mapping / sequence of arrays.  Planes the decode did not ask for are absent (empty)."""
    if hasattr(source, "getDecodingReturnValues"):
        source = source.getDecodingReturnValues()
    if isinstance(source, dict):
        items = list(source.items())
    elif isinstance(source, (list, tuple)):
        items = list(zip(SUM_PLANES, source))
    else:
        items = [(n, getattr(source, n)) for n in SUM_PLANES]
    out = {}
    for name, a in items:
        a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
        if a.size:
            out[name] = a
    return out


def reduce_sums(source, dist=None, rank: int = 0, world: int = 1, device="cpu", order: str = "rank",
                force_collective: bool = False):
    """Sum-over-pairs mode across ranks (ASMC mode: ``HMM.decodeAll(jobs, jobInd)`` with jobs = world, jobInd = rank + 1
    -- the reference's own decomposition of a decode into jobs, HMM.cpp:310-321 -- leaves every rank the posterior sums
    of ITS pairs, HMM.cpp:1044-1085).  The reference merges the jobs' matrices afterwards, job after job, in fp32
    (TOOLS/MERGE_POSTERIORS PosteriorMerger.java:121-204: ``sum[r][c] += job's value`` in job order); this is that merge
    over the process group:

    * ``order="rank"`` (default): every rank's planes are gathered to rank 0 (one ``gather`` of the stacked planes: RCCL
      over xGMI with backend "nccl" and ``device="cuda"``, gloo in CPU tests) and added there ONE RANK AFTER THE OTHER,
      starting from zeros -- ``((0 + P_0) + P_1) + ...``, the reference's merge order, the way the library adds a
      launch's batch planes to its accumulator (``add_planes_in_order_kernel``): the result is the same bits whatever
      the transport.  Against ONE process that decodes all the pairs the sums differ by fp32 re-association at the
      shard joins only: each rank's partial is the exact sequential sum of its batches, so the difference is bounded by
      (world - 1) roundings of the running total per entry -- relative 6e-8 x (world - 1), tests use 1e-6.  Returns a
      dict {plane name: [sites][states] float32} on rank 0, None on the other ranks.
    * ``order="allreduce"``: one ``all_reduce(SUM)`` -- every rank gets the total, but the order of the additions is the
      collective's (ring / tree): RE-ASSOCIATING, not reproducible bit for bit across world sizes or transports.

    A group of one rank returns its own planes without a collective unless ``force_collective`` (the one-rank rehearsal
    of the RCCL leg)."""
    if order not in ("rank", "allreduce"):
        raise ValueError("order is 'rank' or 'allreduce'")
    planes = _sum_planes(source)
    names = [n for n in SUM_PLANES if n in planes] + sorted(n for n in planes if n not in SUM_PLANES)
    if dist is None or (world == 1 and not force_collective):
        return {n: planes[n].copy() for n in names}
    import torch

    shapes = {planes[n].shape for n in names}
    if len(shapes) > 1:
        raise ValueError(f"planes of different shapes: {shapes}")
    # every rank must bring the same planes (a rank whose shard is empty still holds zero-filled matrices)
    have = torch.tensor([sum(1 << i for i, n in enumerate(SUM_PLANES) if n in planes), len(names)], device=device,
                        dtype=torch.int64)
    every = [torch.zeros_like(have) for _ in range(world)]
    dist.all_gather(every, have)
    if any(not torch.equal(e.cpu(), have.cpu()) for e in every):
        raise ValueError("the ranks hold different posterior-sum planes")
    if not names:
        return {} if (rank == 0 or order == "allreduce") else None
    stack = torch.from_numpy(np.stack([planes[n] for n in names])).to(device)
    if order == "allreduce":
        dist.all_reduce(stack)  # (SUM)
        total = stack.cpu().numpy()
        return {n: total[i] for i, n in enumerate(names)}
    bucket = [torch.empty_like(stack) for _ in range(world)] if rank == 0 else None
    dist.gather(stack, bucket, dst=0)
    if rank != 0:
        return None
    acc = np.zeros(stack.shape, np.float32)
    for r in range(world):  # rank order = job order = the reference's merge order
        acc = acc + bucket[r].cpu().numpy()
    return {n: acc[i] for i, n in enumerate(names)}
