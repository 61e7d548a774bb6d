#!/usr/bin/env python3
"""bench.py -- pairs/s of the pairwise HMM decode path on MI355X, against its HBM roofline.

Workload (BASELINE.json configs[1]): synthetic 1000 haplotypes x 50000 sites, 69 states, all
499500 haplotype pairs, FastSMC-mode output (IBD segments + posterior-mean / MAP ages, no hashing),
one GPU.  A "step" = one decode of the whole pair list with every input (model tables, packed
haplotypes, work list) already resident in HBM; the step ends when the ordered IBD records are back
on the host.  With N > 1 (one process per GPU under torch.distributed.run) every rank decodes the
full pair list of its own synthetic cohort (seed + rank) -- weak scaling, no data-path collective --
and the records are gathered to rank 0 over RCCL at the end of each step.

Prints ONE JSON line (see the contract in the task description) with `roofline` and `cpu_baseline`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # B/s, MI355X spec (MI355X_MICROARCH.md); ~6.3e12 is the measured achievable


def measured_traffic(n_hap: int, n_sites: int, K: int, beta_stride: int):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same command
    (profiles/r01c_traffic.json for beta stride 2, profiles/r01_traffic.json for stride 1: separate --pmc FETCH_SIZE /
    WRITE_SIZE runs, gfx950 FETCH_SIZE x2 correction).  PMC collection cannot run inside the timed process, so the
    figure is the committed measurement; it is reported only for the workload and kernel it was measured on."""
    name = {1: "r01_traffic.json", 2: "r01c_traffic.json"}.get(beta_stride)
    if (n_hap, n_sites, K) != (1000, 50000, 69) or name is None:
        return None
    path = os.path.join(ROOT, "profiles", name)
    try:
        return float(json.load(open(path))["hbm_bytes_per_launch"])
    except Exception:
        return None


def build_problem(n_hap: int, n_sites: int, K: int, seed: int):
    """Synthetic cohort + model, prepared by the product's own host code (C++ Data/HMM constructors)."""
    from fastsmc_amd import api, synth

    tables = synth.make_model_tables(K)
    haps = synth.make_haps(n_hap, n_sites, seed=seed)
    data = api.Data.from_arrays(haps.alleles, haps.bp, haps.cm, True, True)
    dq = api.decoding_quantities_from_tables(tables)
    p = api.DecodingParams()
    # FastSMC defaults (DecodingParams.cpp:56-73) except hashing
    p.FastSMC = True
    p.foldData = True
    p.usingCSFS = True
    p.batchSize = 32
    p.time = 50
    p.noConditionalAgeEstimates = True
    p.doPerPairPosteriorMean = True
    p.doPerPairMAP = True
    p.outputIbdSegmentLength = True
    p.useKnownSeed = True
    p.hashing = False
    hmm = api.HMM(data, dq, p)
    pm = api.PreparedModelView(hmm.preparedModel())
    bits = data.packed_bits()
    return pm, bits, haps, tables


def all_pairs(n_ind: int) -> np.ndarray:
    """Pair order of HMM::decodeAll (HMM.cpp:325-357), vectorised: rows (hapA, hapB)."""
    out = []
    for i in range(n_ind):
        if i:
            j = np.repeat(np.arange(i, dtype=np.uint32), 4)
            i_hap = np.tile(np.array([0, 0, 1, 1], np.uint32), i)
            j_hap = np.tile(np.array([0, 1, 0, 1], np.uint32), i)
            out.append(np.stack([2 * j + j_hap, np.full(4 * i, 2 * i, np.uint32) + i_hap], axis=1))
        out.append(np.array([[2 * i, 2 * i + 1]], np.uint32))
    return np.concatenate(out).astype(np.uint32)


def cpu_baseline(pm, haps, n_pairs_sample: int, pairs: np.ndarray) -> dict:
    """The oracle (C restatement of the reference's NO_SSE path; here its -O3 -mavx2 build, which tests check is
    bit-identical to the checker build) on the first pairs of the same work list, same sites, reference batch size
    32, one batch per host thread on every core this process may use.  Reported baseline only."""
    from fastsmc_amd import synth
    from oracle import oracle as O

    _, _, flipped = synth.fold_and_pack(haps.alleles)
    folded = np.where(flipped[None, :], 1 - haps.alleles, haps.alleles).astype(np.uint8)
    model = O.PreparedModel(K=pm.K, S=pm.S, pi=pm.pi, col_ratios=pm.col_ratios, exp_times=pm.exp_times, D=pm.D,
                            B=pm.B, U=pm.U, RR=pm.RR, step_row=pm.step_row, e1=pm.e1, e0m1=pm.e0m1, e2m0=pm.e2m0,
                            gen=np.zeros(pm.S, np.float32), phys=np.zeros(pm.S, np.int32),
                            state_threshold=int(pm.state_threshold), age_threshold=int(pm.age_threshold),
                            probability_threshold=np.float32(pm.probability_threshold))
    # one GPU's share of the host: at most 16 cores (each thread also holds two S x K x 32 float buffers)
    cores = max(1, min(len(os.sched_getaffinity(0)), os.cpu_count() or 1, 16))
    if n_pairs_sample <= 0:  # automatic: four batches of 32 pairs per core
        n_pairs_sample = 4 * 32 * cores
    sample = [tuple(int(x) for x in pr) for pr in pairs[:n_pairs_sample]]
    O.select_build("avx2")
    try:
        t0 = time.perf_counter()
        recs = O.decode_pairs_ibd(model, folded, sample, batch_size=32, threads=cores)
        dt = time.perf_counter() - t0
    finally:
        O.select_build("ref")
    return {"value": len(sample) / dt, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"first {len(sample)} pairs of the same work list x {pm.S} sites (batches of 32, one batch per "
                      f"thread, {cores} threads), {dt:.1f} s wall, {len(recs)} IBD records; oracle/hmm_oracle.c, "
                      f"gcc -O3 -mavx2 -ffp-contract=off"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--haps", type=int, default=1000)
    ap.add_argument("--sites", type=int, default=50000)
    ap.add_argument("--states", type=int, default=69)
    ap.add_argument("--chunk-sites", type=int, default=0, help="sites between beta checkpoints (0 = automatic)")
    ap.add_argument("--beta-stride", type=int, default=0,
                    help="1 = every beta row through HBM, 2 = every second row (others recomputed), 0 = automatic")
    ap.add_argument("--flags", type=int, default=-1, help="FSMC_WANT_* bits (default: mean + MAP ages)")
    ap.add_argument("--diag-same-row", action="store_true",
                    help="diagnostic, NOT a result: every site uses the same transition-table row (scalar-cache hits)")
    ap.add_argument("--ws-frac", type=float, default=0.0, help="workspace cap as a fraction of HBM (0 = default)")
    ap.add_argument("--cpu-pairs", type=int, default=-1,
                    help="pairs in the cpu_baseline sample (0 = skip, -1 = automatic: 128 per host core)")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist  # backend "nccl" is RCCL on ROCm

        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from fastsmc_amd import capi

    pm, bits, haps, _ = build_problem(args.haps, args.sites, args.states, seed=1234 + rank)
    pairs = all_pairs(args.haps // 2)
    n_pairs = int(pairs.shape[0])
    groups = capi.whole_sequence_groups(n_pairs, pm.S, batch=64)

    if args.diag_same_row:
        pm.step_row = np.full_like(pm.step_row, pm.step_row[1])
    ctx = capi.Context(local_rank)
    model = ctx.create_model(pm)
    ctx.upload_haps(bits, pm.S)
    ctx.upload_worklist(pairs.view(capi.PAIR_DTYPE).reshape(-1), groups)
    if args.chunk_sites:
        ctx.set_chunk_sites(args.chunk_sites)
    if args.ws_frac:
        ctx.set_workspace_limit(int(args.ws_frac * ctx.info()["hbm_bytes"]))
    if args.beta_stride:
        ctx.set_beta_stride(args.beta_stride)
    flags = (capi.FSMC_WANT_MEAN | capi.FSMC_WANT_MAP) if args.flags < 0 else args.flags

    from fastsmc_amd.dist import gather_ibd_records

    def gather_records(rec: np.ndarray):
        """The path's only exchange: variable-length IBD records to rank 0 (counts, then padded payloads)."""
        total, _ = gather_ibd_records(rec, rank * n_pairs, dist, rank, world, device="cuda")
        return total

    def step():
        ctx.decode_ibd_launch(model, flags)
        rec = ctx.decode_ibd_fetch()
        return rec, gather_records(rec)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    kernel_ms = []
    barrier()
    t0 = time.perf_counter()
    n_rec = 0
    for _ in range(args.steps):
        _, n_rec = step()
        kernel_ms.append(ctx.last_kernel_ms())
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    info = ctx.info()
    phase = ctx.phase_cycles()
    if rank == 0:
        total_pairs = n_pairs * world * args.steps
        value = total_pairs / elapsed
        k_s = float(np.mean(kernel_ms)) / 1e3
        bytes_per_pair_site = 8 * pm.K + 0.25  # beta row written + read once, + 2 genotype bits
        algo_bytes = n_pairs * pm.S * bytes_per_pair_site
        achieved = algo_bytes / k_s
        out = {
            "metric": "haplotype-pairs decoded/sec (whole node) + GB/s vs HBM roofline, 69-state HMM",
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"synthetic {args.haps} haplotypes x {args.sites} sites, K={pm.K}, all "
                                   f"{n_pairs} pairs per GPU, FastSMC-mode IBD + posterior-mean/MAP ages, no hashing",
                       "pair_sites_per_s": value * pm.S, "ibd_records_per_step": n_rec,
                       "resident_waves": info["n_slots"], "n_cu": info["n_cu"],
                       "chunk_sites": info["chunk_sites"], "chunks_per_window": info["max_chunks"],
                       "beta_stride": ctx.last_beta_stride(),
                       **({"DIAGNOSTIC_same_row": True} if args.diag_same_row else {}),
                       **({"phase_cycles": [int(x) for x in phase]} if phase.any() else {})},
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK, "traffic": measured_traffic(args.haps, args.sites, pm.K, ctx.last_beta_stride()),
                         "kernel_ms": 1e3 * k_s, "algorithmic_bytes_per_launch": algo_bytes},
        }
        if args.cpu_pairs != 0 and world == 1:
            out["cpu_baseline"] = cpu_baseline(pm, haps, args.cpu_pairs, pairs)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
