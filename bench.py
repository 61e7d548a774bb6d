#!/usr/bin/env python3
"""bench.py -- pairs/s of the pairwise HMM decode path on MI355X, against its HBM roofline.

N = 1 (default; BASELINE.json configs[1], "C2"): synthetic 1000 haplotypes x 50000 sites, 69 states, all 499500
haplotype pairs, FastSMC-mode output (IBD segments + posterior-mean / MAP ages, no hashing), one GPU.  Behind the timed
region the same line carries one step each of the other single-GPU configurations (`config.other_workloads`: the
FASTSMC_EXAMPLE shape C1 as IBD decode and as sum over pairs, and the 256-state configuration C4) and the C2 step of a
context that runs on the library's default workspace policy (`config.library_default_workspace`).

N > 1 (one process per GPU under torch.distributed.run; BASELINE.json configs[2], "C3"): STRONG scaling of ONE
problem -- synthetic 10000 haplotypes x 100000 sites, 69 states, a fixed seeded sub-list of 5 * 2^19 = 2 621 440 of the 49 995 000
pairs in the reference's enumeration order.  The work list is cut into contiguous shards of equal pair-site weight
(whole 64-pair groups: the reference's own job decomposition, HMM.cpp:310-321), every rank holds the model and the
packed haplotypes, there is no collective on the data path, and the variable-length IBD records are gathered to
rank 0 over RCCL at the end of each step.  `--workload c3` runs the same list on one GPU (the N = 1 point of the
strong-scaling curve; profiles/ keeps that measurement and the N > 1 line quotes it when it is of this build).
The cohort is synthesised and prepared ONCE per node (local rank 0; the other ranks map its arrays from a cache file)
and every rank's host threads are capped at cores / ranks.

A "step" = one decode of the whole pair list with every input (model tables, packed haplotypes, work list) already
resident in HBM; the step ends when the ordered IBD records are back on the host of rank 0.

Prints ONE JSON line (see the contract in the task description) with `roofline` and, at N = 1, `cpu_baseline`.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # B/s, MI355X spec (MI355X_MICROARCH.md); ~6.3e12 is the measured achievable
METRIC = "haplotype-pairs decoded/sec (whole node) + GB/s vs HBM roofline, 69-state HMM"
# 40 960 groups of 64 pairs = 20 rounds of one GPU's 2048 resident waves; at N = 8 a GPU's share is 2.5 rounds -- NOT a
# whole number (a list of 2^20 pairs gave every GPU exactly one round at N = 8: no tail, a flattering curve)
C3_PAIRS = 5 << 19
C3_SEED = 20260
PROFILE_ROUND = "r05"  # profiles/<round>_traffic.json, <round>_c3_n1.json: the committed measurements of this build


def lib_hash() -> str:
    from fastsmc_amd import build

    return build.hip_source_hash()


def committed_measurement(name: str):
    """A measurement of THIS build kept under profiles/ (PMC traffic, the one-GPU point of the strong-scaling
    curve): the file carries the hash of the HIP sources it was taken with and is ignored when they have changed."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None
    return d if d.get("lib_hash") == lib_hash() else None


def measured_traffic(workload_key: str, beta_stride: int):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes of this same command (separate
    --pmc FETCH_SIZE / WRITE_SIZE runs, gfx950 FETCH_SIZE x2 correction; tools/profile_bench.sh +
    tools/stamp_traffic.py).  PMC collection cannot run inside the timed process, so the figure is the committed
    measurement -- reported only for the workload, stride and library build it was measured on, else null."""
    d = committed_measurement(f"{PROFILE_ROUND}_traffic.json")
    if not d or d.get("workload_key") != workload_key or d.get("beta_stride") != beta_stride:
        return None
    return float(d["hbm_bytes_per_launch"])


def fastsmc_params():
    from fastsmc_amd import api

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
    return p


def build_problem(n_hap: int, n_sites: int, K: int, seed: int, blocked: bool = False):
    """Synthetic cohort + model, prepared by the product's own host code (C++ Data/HMM constructors)."""
    from fastsmc_amd import api, synth

    tables = synth.make_model_tables(K)
    haps = (synth.make_haps_blocked if blocked else synth.make_haps)(n_hap, n_sites, seed=seed)
    data = api.Data.from_arrays(haps.alleles, haps.bp, haps.cm, True, True)
    dq = api.decoding_quantities_from_tables(tables)
    hmm = api.HMM(data, dq, fastsmc_params())
    pm = api.PreparedModelView(hmm.preparedModel())
    bits = data.packed_bits()
    return pm, bits, haps, tables


def _cache_key(*parts) -> str:
    """Everything the prepared arrays depend on: the shape and seed, the synthesiser and the host preparation code."""
    h = hashlib.sha256(repr(parts).encode())
    host = os.path.join(ROOT, "fastsmc_amd", "csrc", "host")
    for path in [os.path.join(ROOT, "fastsmc_amd", "synth.py")] + sorted(
            os.path.join(host, f) for f in os.listdir(host) if f.endswith((".cpp", ".hpp"))):
        h.update(open(path, "rb").read())
    return h.hexdigest()[:20]


def node_cached_problem(n_hap: int, n_sites: int, K: int, seed: int, local_rank: int, timeout_s: float = 300.0):
    """The C3 cohort (10 000 haplotypes x 100 000 sites: half a minute of synthesis and preparation, 1 GB of alleles
    on the way) is built ONCE per node: local rank 0 builds it and writes the prepared arrays -- the model view and the
    packed haplotypes, 0.3 GB -- to a cache file (atomic rename); the other ranks wait for the file and map it.  A
    second run on the same box (the driver's N = 2, 4, 8 in a row) finds the file.  FSMC_BENCH_CACHE names the
    directory (default: $XDG_CACHE_HOME/fsmc_bench, else a per-user directory of mode 0700 under the system's temporary
    directory); the file carries a checksum of its arrays and is ignored (and rebuilt) when it does not verify or the
    directory is writable by others; a rank that cannot load it builds the cohort itself, and when local rank 0 cannot
    write it leaves a `.failed` marker so that the others do not wait."""
    from fastsmc_amd import api

    # a per-user directory nobody else can write to: the arrays in it are trusted (they shape the C3 line)
    base = os.environ.get("FSMC_BENCH_CACHE") or os.path.join(
        os.environ.get("XDG_CACHE_HOME") or os.path.join(tempfile.gettempdir(), f"fsmc_cache_{os.getuid()}"), "fsmc_bench")
    cache_dir = base
    path = os.path.join(cache_dir, f"cohort_{_cache_key(n_hap, n_sites, K, seed, 'blocked')}.npz")
    failed = path + ".failed"

    def digest(arrays: dict) -> str:
        h = hashlib.sha256()
        for k in sorted(arrays):
            a = np.ascontiguousarray(arrays[k])
            h.update(k.encode())
            h.update(str(a.dtype).encode())
            h.update(repr(a.shape).encode())
            h.update(a.tobytes())
        return h.hexdigest()

    def load():
        st = os.stat(cache_dir)
        if st.st_uid != os.getuid() or (st.st_mode & 0o022):
            raise OSError("cache directory is not this user's own")
        with np.load(path) as z:
            arrays = {k: z[k] for k in z.files if k != "__sha256"}
            want = str(z["__sha256"])
        if digest(arrays) != want:
            raise ValueError("cohort cache: checksum mismatch")
        bits = arrays.pop("__bits")
        d = {k: (v if v.ndim else v.item()) for k, v in arrays.items()}
        return api.PreparedModelView(d), bits

    def build_here():
        return build_problem(n_hap, n_sites, K, seed=seed, blocked=True)[:2]

    if local_rank != 0:
        # wait for local rank 0's file (or for its word that there will be none); a file that does not load -- stale,
        # truncated, tampered with -- is not worth a crash: build the cohort here
        t0 = time.time()
        while not os.path.exists(path):
            if os.path.exists(failed) or time.time() - t0 > timeout_s:
                return build_here()
            time.sleep(0.2)
        time.sleep(0.05)
        try:
            return load()
        except Exception:
            return build_here()
    if os.path.exists(path):
        try:
            return load()
        except Exception:
            pass  # (a truncated or foreign file: rebuild and replace it)
    pm, bits = build_here()
    try:
        os.makedirs(cache_dir, mode=0o700, exist_ok=True)
        os.chmod(cache_dir, 0o700)
        if os.path.exists(failed):
            os.remove(failed)
        arrays = {"__bits": np.asarray(bits), **{k: np.asarray(v) for k, v in pm.__dict__.items()}}
        tmp = f"{path}.{os.getpid()}.tmp.npz"
        np.savez(tmp, __sha256=np.array(digest(arrays)), **arrays)
        os.replace(tmp, path)
    except OSError:
        try:  # the other ranks stop waiting
            os.makedirs(cache_dir, exist_ok=True)
            open(failed, "w").close()
        except OSError:
            pass
    return pm, bits


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


def folded_rows_from_bits(bits: np.ndarray, rows, n_sites: int) -> np.ndarray:
    """Folded alleles [len(rows)][n_sites] uint8 of the given haplotype rows, unpacked from the packed matrix the GPU
    decodes from ([hap][ceil(S/64)] little-endian u64, site s at bit s % 64)."""
    words = np.ascontiguousarray(np.asarray(bits)[np.asarray(rows)]).view(np.uint8)
    return np.unpackbits(words, axis=1, bitorder="little")[:, :n_sites]


def cpu_baseline(pm, bits: np.ndarray, n_pairs_sample: int, pairs: np.ndarray) -> dict:
    """The oracle (C restatement of the reference's NO_SSE path; here its -O3 -mavx2 build, which tests check is
    bit-identical to the checker build) on the first pairs of the same work list, same sites, reference batch size
    32, one batch per host thread on every core this process may use.  Reported baseline only.  The folded alleles
    of the sampled haplotypes come out of the packed matrix the GPU decodes from."""
    from oracle import oracle as O

    model = O.PreparedModel(K=pm.K, S=pm.S, pi=pm.pi, col_ratios=pm.col_ratios, exp_times=pm.exp_times, D=pm.D,
                            B=pm.B, U=pm.U, RR=pm.RR, step_row=pm.step_row, e1=pm.e1, e0m1=pm.e0m1, e2m0=pm.e2m0,
                            gen=np.zeros(pm.S, np.float32), phys=np.zeros(pm.S, np.int32),
                            state_threshold=int(pm.state_threshold), age_threshold=int(pm.age_threshold),
                            probability_threshold=np.float32(pm.probability_threshold))
    # one GPU's share of the host: at most 16 cores (each thread also holds two S x K x 32 float buffers)
    cores = max(1, min(len(os.sched_getaffinity(0)), os.cpu_count() or 1, 16))
    if n_pairs_sample <= 0:  # automatic: four batches of 32 pairs per core
        n_pairs_sample = 4 * 32 * cores
    sample_pairs = np.asarray(pairs[:n_pairs_sample], dtype=np.int64)
    used, inverse = np.unique(sample_pairs, return_inverse=True)
    folded = folded_rows_from_bits(bits, used, pm.S)
    sample = [tuple(int(x) for x in pr) for pr in inverse.reshape(sample_pairs.shape)]
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


def algorithmic_bytes(n_pairs: int, S: int, K: int) -> float:
    """beta row written + read once, + 2 genotype bits, per pair-site."""
    return float(n_pairs) * S * (8 * K + 0.25)


def _timed(fn, reps: int, ctx):
    """One untimed pass, then `reps` timed ones: (mean kernel ms by the library's HIP events, mean wall s of the call)."""
    ms, wall = [], []
    for it in range(1 + reps):
        t0 = time.perf_counter()
        fn()
        if it:
            wall.append(time.perf_counter() - t0)
            ms.append(ctx.last_kernel_ms())
    return float(np.mean(ms)), float(np.mean(wall))


def consumer_line(name: str, shape: str, algo: float, k_ms: float, wall_s: float, n_pairs: int, ctx, extra: dict) -> dict:
    return {"workload": name, "shape": shape, "kernel_ms": k_ms, "call_s": wall_s, "frac": algo / (k_ms / 1e3) / HBM_PEAK,
            "pairs_per_s_kernel": n_pairs / (k_ms / 1e3), "kernel_member": ctx.last_kernel(), "lib_hash": lib_hash(),
            **extra}


def c2_consumers(ctx, capi, model, pm, pairs: np.ndarray, out_budget_bytes: float = 16e9) -> list:
    """The other posterior consumers on the headline problem, which is resident (model, haplotypes, workspace): the sum
    over pairs of ALL pairs (HMM.cpp:1044-1085; output [sites][states], 13.8 MB) and the per-pair posterior mean + MAP
    rows (HMM.cpp:1378-1409; 8 B per pair-site of output, which is what bounds the list: the first pairs whose rows
    fit `out_budget_bytes`).  `frac` is on the same byte model as the headline (8K + 0.25 B per pair-site); the
    consumer's own output bytes are stated beside it.  `call_s` is the whole C-ABI call (kernel + the copy of the
    output to the host)."""
    out = []
    n_all = int(pairs.shape[0])
    shape = f"{pm.S} sites, K={pm.K}"
    # sums over pairs, every group a batch of 64 (the plane-ordered accumulation of DESIGN.md section 1, a9)
    k_ms, wall = _timed(lambda: ctx.decode_sums(model), 1, ctx)
    out.append(consumer_line("c2_sums", f"all {n_all} pairs x {shape}", algorithmic_bytes(n_all, pm.S, pm.K), k_ms, wall,
                             n_all, ctx, {"consumer_output_bytes": 4.0 * pm.S * pm.K, "steps": 1}))
    # per-pair mean + MAP rows
    n_pp = int(min(n_all, out_budget_bytes / (8.0 * pm.S))) // 64 * 64
    ctx.upload_worklist(pairs[:n_pp].view(capi.PAIR_DTYPE).reshape(-1), capi.whole_sequence_groups(n_pp, pm.S, batch=64))
    k_ms, wall = _timed(lambda: ctx.decode_per_pair(model, pm.exp_times), 1, ctx)
    out.append(consumer_line("c2_per_pair", f"first {n_pp} pairs ({n_pp // 64} groups on {ctx.info()['n_slots']} resident "
                             f"waves: the rows of more pairs do not fit {out_budget_bytes / 1e9:.0f} GB) x {shape}",
                             algorithmic_bytes(n_pp, pm.S, pm.K), k_ms, wall, n_pp, ctx,
                             {"consumer_output_bytes": 8.0 * n_pp * pm.S, "steps": 1}))
    return out


def other_workloads(ctx, capi, flags: int, budget_s: float) -> list:
    """One step each of the other single-GPU configurations on the context of the headline run (its workspace is
    allocated already): C1 = the FASTSMC_EXAMPLE shape (300 haplotypes x 6760 sites, K = 69, all 44 850 pairs) as IBD
    decode, as sum over pairs (the reference's published ASMC regression job, time_regression.py), as per-pair mean +
    MAP rows and -- its first 8192 pairs: 4K B of output per pair-site -- as full posterior dump (ASMC.decodePairs,
    ASMC.cpp:80-128), with the CPU port timed on the same shape; a 600-state model on the 600 x 3000 list (179 700 pairs:
    the wave-group kernel's eight waves of 80 states, no landing zones); C4 = 256 states x 200 000-site windows on a
    256-haplotype sub-cohort (32 640 pairs: 510 groups = ONE round of the 512 resident workgroups) and on a 512-haplotype
    one (130 816 pairs: 2044 groups = four rounds).  Not part of `value`."""
    out = []
    t_begin = time.perf_counter()
    for name, (n_hap, n_sites, K), modes in (("c1", (300, 6760, 69), ("ibd", "sums", "per_pair", "dump")),
                                             ("k600_list", (600, 3000, 600), ("ibd",)),
                                             ("c4", (256, 200000, 256), ("ibd",)),
                                             ("c4_four_rounds", (512, 200000, 256), ("ibd",))):
        if time.perf_counter() - t_begin > budget_s:
            out.append({"workload": name, "skipped": "time budget"})
            continue
        pm, bits, _, _ = build_problem(n_hap, n_sites, K, seed=1234)
        pairs = all_pairs(n_hap // 2)
        model = ctx.create_model(pm)
        ctx.upload_haps(bits, pm.S)
        n_all = int(pairs.shape[0])
        shape = f"{n_hap} haplotypes x {n_sites} sites, K={pm.K}"

        def use(n):
            ctx.upload_worklist(pairs[:n].view(capi.PAIR_DTYPE).reshape(-1), capi.whole_sequence_groups(n, pm.S, batch=64))

        for mode in modes:
            reps = 3 if name == "c1" else 1
            n = min(n_all, 8192) if mode == "dump" else n_all
            use(n)
            extra = {"steps": reps}
            if mode == "ibd":
                n_rec = [0]

                def run():
                    ctx.decode_ibd_launch(model, flags)
                    n_rec[0] = int(ctx.decode_ibd_fetch().size)
                k_ms, wall = _timed(run, reps, ctx) if name != "c4_four_rounds" else _timed(run, 1, ctx)
                extra.update(ibd_records=n_rec[0], beta_stride=ctx.last_beta_stride(),
                             groups=(n + 63) // 64, resident_waves_or_workgroups=ctx.info()["n_slots"])
            elif mode == "sums":
                k_ms, wall = _timed(lambda: ctx.decode_sums(model), reps, ctx)
                extra["consumer_output_bytes"] = 4.0 * pm.S * pm.K
            elif mode == "per_pair":
                k_ms, wall = _timed(lambda: ctx.decode_per_pair(model, pm.exp_times), reps, ctx)
                extra["consumer_output_bytes"] = 8.0 * n * pm.S
            else:
                k_ms, wall = _timed(lambda: ctx.decode_posteriors(model), reps, ctx)
                extra["consumer_output_bytes"] = 4.0 * pm.K * 64.0 * ((n + 63) // 64) * pm.S
                extra["groups"] = (n + 63) // 64
            out.append(consumer_line(f"{name}_{mode}", f"{shape}, {'all' if n == n_all else 'first'} {n} pairs",
                                     algorithmic_bytes(n, pm.S, pm.K), k_ms, wall, n, ctx, extra))
        if name == "c1":
            # north_star's ">= 10x the host CPU on FASTSMC_EXAMPLE-shaped input": the CPU port on this shape (IBD mode),
            # 16 batches of 32 pairs per core
            cores = max(1, min(len(os.sched_getaffinity(0)), os.cpu_count() or 1, 16))
            cb = cpu_baseline(pm, bits, min(n_all, 16 * 32 * cores), pairs)
            ibd = next(o for o in out if o["workload"] == "c1_ibd")
            out.append({"workload": "c1_cpu_baseline", **cb, "gpu_kernel_over_cpu": ibd["pairs_per_s_kernel"] / cb["value"]})
        model.close()
    return out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=("auto", "c1", "c2", "c3", "c4"), default="auto",
                    help="auto: c2 at N = 1, c3 (strong scaling of one sharded list) at N > 1; c1: the FASTSMC_EXAMPLE "
                         "shape (300 haplotypes x 6760 sites, all 44850 pairs); c4: 256 states, 200000-site windows "
                         "(a 256-haplotype sub-cohort of config 4: 32640 pairs)")
    ap.add_argument("--mode", choices=("ibd", "sums"), default="ibd",
                    help="ibd: FastSMC-mode IBD decode (the headline); sums: ASMC-mode sum of the posteriors over pairs "
                         "(HMM.cpp:1044-1085) of ONE cohort (workload c1 or c2) -- at N > 1 rank r decodes job r + 1 of N "
                         "(the reference's own job ranges, HMM.cpp:310-321) and the planes are merged on rank 0 in rank "
                         "order (fastsmc_amd.dist.reduce_sums: the reference's PosteriorMerger order)")
    ap.add_argument("--haps", type=int, default=0, help="haplotypes (default: the workload's)")
    ap.add_argument("--sites", type=int, default=0, help="sites (default: the workload's)")
    ap.add_argument("--pairs", type=int, default=0, help="c3: pairs in the seeded sub-list (default 5 * 2^19)")
    ap.add_argument("--states", type=int, default=69)
    ap.add_argument("--chunk-sites", type=int, default=0, help="sites between beta checkpoints (0 = automatic)")
    ap.add_argument("--beta-stride", type=int, default=0,
                    help="1 = every beta row through HBM, 2 = every second row (others recomputed), 0 = automatic")
    ap.add_argument("--flags", type=int, default=-1, help="FSMC_WANT_* bits (default: mean + MAP ages)")
    ap.add_argument("--diag-same-row", action="store_true",
                    help="diagnostic, NOT a result: every site uses the same transition-table row (scalar-cache hits)")
    ap.add_argument("--ws-frac", type=float, default=0.8,
                    help="workspace limit as a fraction of HBM, set by the caller like a long job would (a step is a "
                         "slice of one: the library's own default lets a context EARN its workspace over its first "
                         "minutes, DESIGN.md 3.3 -- hipMalloc costs 40 ms per GB); 0 = that default.  The default line "
                         "carries BOTH figures (config.library_default_workspace)")
    ap.add_argument("--resident-chunks", type=int, default=-1,
                    help="chunks of a chunked window whose beta rows stay in the workspace (no rebuild): -1 = as many "
                         "as memory allows, 0 = none")
    ap.add_argument("--cpu-pairs", type=int, default=-1,
                    help="pairs in the cpu_baseline sample (0 = skip, -1 = automatic: 128 per host core)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--force-collective", action="store_true",
                    help="N = 1: bring the process group up all the same (one rank) and send the records through the "
                         "path's collectives -- the rehearsal of the RCCL leg on a one-GPU box")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="skip config.other_workloads / config.library_default_workspace (default N = 1 line only)")
    ap.add_argument("--startup-only", action="store_true",
                    help="build (or map) the problem and the work list, print the start-up time per rank and stop "
                         "before the first GPU call (measures the host start-up where there is no GPU)")
    ap.add_argument("--dump-records", default="", help="rank 0 writes the last step's gathered records to this .npy")
    args = ap.parse_args()
    t_start = time.perf_counter()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if local_world > 1 and "FSMC_HOST_THREADS" not in os.environ:
        # the host's cores are shared by the ranks of the node: the product's start-up threads (reader, transposes,
        # emission preparation) take cores / ranks each
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        os.environ["FSMC_HOST_THREADS"] = str(max(1, cores // local_world))

    import torch

    device = local_rank % max(1, torch.cuda.device_count())  # (rehearsals put several ranks on one card)
    dist = None
    use_group = world > 1 or args.force_collective
    if not args.startup_only:
        torch.cuda.set_device(device)
    if use_group:
        import torch.distributed as dist  # backend "nccl" is RCCL on ROCm

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if args.backend == "nccl" and not args.startup_only:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo" if args.startup_only else args.backend, rank=rank, world_size=world)
    comm_device = "cuda" if (args.backend == "nccl" and not args.startup_only) else "cpu"

    from fastsmc_amd import capi
    from fastsmc_amd.dist import (all_pairs_at, gather_ibd_records, reduce_sums, sample_pair_ordinals,
                                  shard_groups_by_weight)

    workload = args.workload if args.workload != "auto" else ("c2" if world == 1 else "c3")
    if args.mode == "sums":
        # ONE cohort for all ranks; rank r takes job r + 1 of `world` of the reference's enumeration (HMM.cpp:319-321)
        if workload not in ("c1", "c2"):
            workload = "c1"
        shape = {"c1": (300, 6760, args.states), "c2": (1000, 50000, args.states)}[workload]
        n_hap, n_sites, n_states = args.haps or shape[0], args.sites or shape[1], shape[2]
        pm, bits, _, _ = build_problem(n_hap, n_sites, n_states, seed=1234)
        every_pair = all_pairs(n_hap // 2)
        n_total = int(every_pair.shape[0])
        lo, hi = n_total * rank // world, n_total * (rank + 1) // world
        pairs = my_pairs = every_pair[lo:hi]
        desc = (f"synthetic {n_hap} haplotypes x {n_sites} sites, K={pm.K}, all {n_total} pairs in {world} job(s) of the "
                f"reference's enumeration (one per GPU), ASMC-mode sum of posteriors over pairs, merged on rank 0 in "
                f"rank order")
        scaling = "strong" if world > 1 else "n/a"
        workload_key = f"{workload}_sums:{n_hap}x{n_sites}:K{pm.K}"
    elif workload in ("c1", "c2", "c4"):
        # every rank its own cohort (only ever run at N = 1 by the driver; N > 1 here is a weak-scaling rehearsal)
        shape = {"c1": (300, 6760, args.states), "c2": (1000, 50000, args.states), "c4": (256, 200000, 256)}[workload]
        n_hap, n_sites, n_states = args.haps or shape[0], args.sites or shape[1], shape[2]
        pm, bits, _, _ = build_problem(n_hap, n_sites, n_states, seed=1234 + rank)
        pairs = all_pairs(n_hap // 2)
        n_total = int(pairs.shape[0]) * world
        lo, my_pairs = rank * int(pairs.shape[0]), pairs
        desc = (f"synthetic {n_hap} haplotypes x {n_sites} sites, K={pm.K}, all {pairs.shape[0]} pairs per GPU, "
                f"FastSMC-mode IBD + posterior-mean/MAP ages, no hashing")
        scaling = "weak" if world > 1 else "n/a"  # (one GPU: there is nothing to scale)
        workload_key = f"{workload}:{n_hap}x{n_sites}:K{pm.K}"
    else:
        # ONE problem for all ranks: same seed everywhere, the work list sharded by pair-site weight
        n_hap, n_sites = args.haps or 10000, args.sites or 100000
        n_list = args.pairs or C3_PAIRS
        pm, bits = node_cached_problem(n_hap, n_sites, args.states, 1234, local_rank)
        ordinals = sample_pair_ordinals(n_hap // 2, n_list, C3_SEED)
        n_total = int(ordinals.size)
        all_groups = capi.whole_sequence_groups(n_total, pm.S, batch=64)
        weights = all_groups["n_pairs"].astype(np.float64) * (all_groups["to"] - all_groups["from"])
        g_lo, g_hi = shard_groups_by_weight(weights, rank, world)
        lo = int(all_groups["first_pair"][g_lo]) if g_lo < all_groups.size else n_total
        hi = int(all_groups["first_pair"][g_hi]) if g_hi < all_groups.size else n_total
        pairs = my_pairs = all_pairs_at(ordinals[lo:hi])
        tot = 2 * (n_hap // 2) ** 2 - n_hap // 2
        desc = (f"synthetic {n_hap} haplotypes x {n_sites} sites, K={pm.K}, a seeded sub-list of {n_total} of the "
                f"{tot} pairs (enumeration order), sharded over {world} GPU(s) by pair-site weight, FastSMC-mode IBD "
                f"+ posterior-mean/MAP ages, no hashing")
        scaling = "strong"  # (total work is fixed whatever N: --workload c3 at N = 1 is the first point of the curve)
        workload_key = f"c3:{n_hap}x{n_sites}:K{pm.K}:{n_total}"
    n_mine = int(my_pairs.shape[0])
    groups = capi.whole_sequence_groups(n_mine, pm.S, batch=64)
    startup_s = time.perf_counter() - t_start

    if args.startup_only:
        every = [startup_s]
        if dist is not None:
            t = torch.tensor([startup_s], dtype=torch.float64)
            got = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(got, t)
            every = [float(x.item()) for x in got]
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"startup_only": True, "workload": desc, "n_ranks": world, "startup_s_per_rank": every,
                              "host_threads_per_rank": os.environ.get("FSMC_HOST_THREADS", "default"),
                              "pairs_of_rank_0": n_mine}))
        return

    if args.diag_same_row:
        pm.step_row = np.full_like(pm.step_row, pm.step_row[1])
    ctx = capi.Context(device)
    model = ctx.create_model(pm)
    ctx.upload_haps(bits, pm.S)
    ctx.upload_worklist(my_pairs.view(capi.PAIR_DTYPE).reshape(-1), groups)
    if args.chunk_sites:
        ctx.set_chunk_sites(args.chunk_sites)
    if args.ws_frac:
        ctx.set_workspace_limit(int(args.ws_frac * ctx.info()["hbm_bytes"]))
    if args.beta_stride:
        ctx.set_beta_stride(args.beta_stride)
    if args.resident_chunks >= 0:
        ctx.set_resident_chunks(args.resident_chunks)
    flags = (capi.FSMC_WANT_MEAN | capi.FSMC_WANT_MAP) if args.flags < 0 else args.flags

    def step():
        """Decode this rank's shard; the path's only exchange: variable-length IBD records to rank 0."""
        if args.mode == "sums":
            plane, _ = ctx.decode_sums(model)
            total = reduce_sums({"sumOverPairs": plane}, dist, rank, world, device=comm_device,
                                force_collective=args.force_collective)
            return (0, None) if total is None else (int(total["sumOverPairs"].shape[0]), total["sumOverPairs"])
        ctx.decode_ibd_launch(model, flags)
        rec = ctx.decode_ibd_fetch()
        return gather_ibd_records(rec, lo, dist, rank, world, device=comm_device,
                                  force_collective=args.force_collective)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    warm_s = []
    for _ in range(args.warmup):
        t0 = time.perf_counter()
        step()
        warm_s.append(time.perf_counter() - t0)
    kernel_ms = []
    barrier()
    t0 = time.perf_counter()
    n_rec, merged = 0, None
    for _ in range(args.steps):
        n_rec, merged = step()
        kernel_ms.append(ctx.last_kernel_ms())
    barrier()
    elapsed = time.perf_counter() - t0
    my_kernel_ms = float(np.mean(kernel_ms))
    per_rank_ms = [my_kernel_ms]
    if dist is not None:
        t = torch.tensor([elapsed, my_kernel_ms], device=comm_device, dtype=torch.float64)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        elapsed = max(float(e[0].item()) for e in every)  # MAX over ranks
        per_rank_ms = [float(e[1].item()) for e in every]

    info = ctx.info()
    phase = ctx.phase_cycles()
    out = None
    if rank == 0:
        if args.dump_records and merged is not None:
            np.save(args.dump_records, merged)
        value = n_total * args.steps / elapsed
        k_s = my_kernel_ms / 1e3
        algo_bytes = algorithmic_bytes(n_mine, pm.S, pm.K)  # the launch this rank times: its own shard
        achieved = algo_bytes / k_s
        traffic = measured_traffic(workload_key, ctx.last_beta_stride())
        out = {
            "metric": METRIC,
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc,
                       "mode": args.mode,
                       **({"sums_checksum": float(np.asarray(merged, np.float64).sum()),
                           "sums_reduction": "none" if dist is None else "rank order on rank 0"}
                          if args.mode == "sums" and merged is not None else {}),
                       "pair_sites_per_s": value * pm.S, "ibd_records_per_step": n_rec,
                       "resident_waves": info["n_slots"], "n_cu": info["n_cu"],
                       "chunk_sites": info["chunk_sites"], "chunks_per_window": info["max_chunks"],
                       "resident_chunks": ctx.last_resident_chunks(),
                       "workspace_limit_frac_of_hbm": args.ws_frac,
                       "first_warmup_step_s": warm_s[0] if warm_s else None,  # (pays the workspace allocation)
                       "beta_stride": ctx.last_beta_stride(), "kernel_member": ctx.last_kernel(),
                       "record_gather": ("none" if dist is None else "rccl" if args.backend == "nccl"
                                         else args.backend),
                       "startup_s": startup_s,
                       "lib_hash": lib_hash(),
                       **({"kernel_ms_per_rank": per_rank_ms,
                           "imbalance_max_over_mean": max(per_rank_ms) / (sum(per_rank_ms) / len(per_rank_ms))}
                          if world > 1 else {}),
                       **({"DIAGNOSTIC_same_row": True} if args.diag_same_row else {}),
                       **({"phase_cycles": [int(x) for x in phase]} if phase.any() else {})},
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK, "frac_algorithmic": achieved / HBM_PEAK,
                         "traffic": traffic,
                         "frac_measured_bytes": (traffic / k_s / HBM_PEAK) if traffic else None,
                         "kernel_ms": 1e3 * k_s, "algorithmic_bytes_per_launch": algo_bytes},
        }
        if scaling == "strong" and world > 1:
            ref = committed_measurement(f"{PROFILE_ROUND}_c3_n1.json")
            if ref and ref.get("workload_key") == workload_key:
                out["config"]["n1_pairs_per_s_same_worklist"] = ref["value"]
                out["config"]["speedup_vs_n1"] = value / ref["value"]
        # the extra measurements ride on the DEFAULT line only: any option that shapes the workload or the plan makes
        # the run a measurement of its own
        default_line = (world == 1 and args.workload == "auto" and args.mode == "ibd" and not args.no_other_workloads
                        and not args.diag_same_row and not args.haps and not args.sites and args.states == 69
                        and args.flags < 0 and not args.chunk_sites and not args.beta_stride
                        and args.resident_chunks < 0 and not args.force_collective)
        if default_line:
            out["config"]["other_workloads"] = (c2_consumers(ctx, capi, model, pm, my_pairs)
                                                + other_workloads(ctx, capi, flags, budget_s=120.0))
        if default_line and args.ws_frac:
            # the same step on the library's own workspace policy (no caller limit: a young context has earned
            # little): what FastSMC.run() gets in its first seconds
            ctx.close()
            ctx = capi.Context(device)
            m2 = ctx.create_model(pm)
            ctx.upload_haps(bits, pm.S)
            ctx.upload_worklist(my_pairs.view(capi.PAIR_DTYPE).reshape(-1), groups)
            ms = []
            for it in range(3):
                ctx.decode_ibd_launch(m2, flags)
                ctx.decode_ibd_fetch()
                if it:
                    ms.append(ctx.last_kernel_ms())
            k2 = float(np.mean(ms)) / 1e3
            out["config"]["library_default_workspace"] = {
                "kernel_ms": 1e3 * k2, "frac": algo_bytes / k2 / HBM_PEAK,
                "resident_chunks": ctx.last_resident_chunks(), "chunk_sites": ctx.info()["chunk_sites"], "steps": 2}
        if args.cpu_pairs != 0 and world == 1:
            out["cpu_baseline"] = cpu_baseline(pm, bits, args.cpu_pairs, pairs)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
