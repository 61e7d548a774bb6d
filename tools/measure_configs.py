#!/usr/bin/env python3
"""Side measurements on one MI355X for the BASELINE.json configs that are not the bench line (reported in DESIGN.md,
never as bench.py's `value`):
  c1_ibd / c1_sums   300 haplotypes x 6760 sites, K = 69, all 44 850 pairs: IBD consumer, and the sum-over-pairs
                     consumer of the reference's own published timing (time_regression.py: 51.97 s on one CPU thread)
  k256 k192 k128 k100  wide models (four waves per group beyond 128 states), reduced C4 shape: 600 haplotypes x 3000
                     sites, all pairs
  hashing            FastSMC.run() end to end with the hashing pre-filter on, C2-sized files (parse + identify + decode +
                     write)
  short              the IBD decode alone on 60 000 hashing-style batches (32 pairs, 320-5504-site windows): the C5 regime
  identify           the identification step alone (fsmc_identify) on the C2 cohort (1000 x 50 000) and on a C3-shaped
                     one (10 000 x 100 000): pair-words/s of the three kernels, with the host restatement beside it
  c5_job             one job window of a biobank-scale run: 16 384 haplotypes x 20 000 sites, hashing on, job 2 of 4
  ingest_c3          host start-up at the C3 shape: Data(params) and HMM(data, params) on a 10 000 x 100 000 .hap.gz
Prints one JSON object per measurement.  Usage: python tools/measure_configs.py [c1 k256 k192 k128 k100 hashing short identify]"""
from __future__ import annotations

import copy
import gzip
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from fastsmc_amd import api, capi, synth  # noqa: E402


def prepared(n_hap, n_sites, K, seed=1234):
    tables = synth.make_model_tables(K)
    haps = synth.make_haps(n_hap, n_sites, seed=seed)
    data = api.Data.from_arrays(haps.alleles, haps.bp, haps.cm, True, True)
    dq = api.decoding_quantities_from_tables(tables)
    p = api.DecodingParams()
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
    return api.PreparedModelView(hmm.preparedModel()), data.packed_bits(), haps, tables


def all_pairs(n_ind):
    sys.path.insert(0, ROOT)
    import bench

    return bench.all_pairs(n_ind)


def long_job_workspace(ctx):
    """Kernel-regime measurements: the workspace limit a long job sets or has earned (bench.py does the same).  Without it a
    context this young keeps its plan small -- hipMalloc costs 40 ms per GB -- and e.g. the C1 windows are
    decoded in chunks: 49.5 instead of 40.3 ms.  The product-path measurements (run_c2, hashing, c5_job) use the default."""
    ctx.set_workspace_limit(int(0.8 * ctx.info()["hbm_bytes"]))


def timed(fn, reps=2):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    return (time.perf_counter() - t0) / reps, out


def c1():
    pm, bits, _, _ = prepared(300, 6760, 69)
    pairs = all_pairs(150)
    ctx = capi.Context(0)
    long_job_workspace(ctx)
    model = ctx.create_model(pm)
    ctx.upload_haps(bits, pm.S)
    pr = pairs.view(capi.PAIR_DTYPE).reshape(-1)
    ctx.upload_worklist(pr, capi.whole_sequence_groups(len(pr), pm.S, batch=64))
    dt, rec = timed(lambda: (ctx.decode_ibd_launch(model), ctx.decode_ibd_fetch())[1])
    print(json.dumps({"config": "c1_ibd", "pairs": len(pr), "sites": pm.S, "K": pm.K, "seconds": dt,
                      "pairs_per_s": len(pr) / dt, "pair_sites_per_s": len(pr) * pm.S / dt, "kernel_ms": ctx.last_kernel_ms(),
                      "records": int(rec.size)}))
    dt, _ = timed(lambda: ctx.decode_sums(model))
    print(json.dumps({"config": "c1_sums", "pairs": len(pr), "sites": pm.S, "K": pm.K, "seconds": dt,
                      "pairs_per_s": len(pr) / dt, "pair_sites_per_s": len(pr) * pm.S / dt, "kernel_ms": ctx.last_kernel_ms(),
                      "reference_published_s": 51.97}))
    ctx.close()


def c1_consumers():
    """The consumers without state across sites on the C1 shape -- sums over all 44 850 pairs (701 groups), per-pair mean /
    MAP rows of all pairs, posterior dump of the first 8192 (128 groups) and 1024 pairs (16 groups: an ASMC.decodePairs
    list) -- with two waves per window (the library's choice for a launch this small, csrc/fsmc_kernels_bidir.h) and on
    the one-wave kernel, interleaved on one box."""
    pm, bits, _, _ = prepared(300, 6760, 69)
    pairs = all_pairs(150)
    ctx = capi.Context(0)
    long_job_workspace(ctx)
    model = ctx.create_model(pm)
    ctx.upload_haps(bits, pm.S)

    def kernel_ms(fn, reps):
        fn()
        ms = []
        for _ in range(reps):
            fn()
            ms.append(ctx.last_kernel_ms())
        return float(np.mean(ms))

    def use(n):
        ctx.upload_worklist(pairs[:n].view(capi.PAIR_DTYPE).reshape(-1), capi.whole_sequence_groups(n, pm.S, batch=64))

    for name, n, fn in (("sums", pairs.shape[0], lambda: ctx.decode_sums(model)),
                        ("sums_00_01_11", pairs.shape[0], lambda: ctx.decode_sums(model, major_minor=True)),
                        ("per_pair", pairs.shape[0], lambda: ctx.decode_per_pair(model, pm.exp_times)),
                        ("dump_8192", 8192, lambda: ctx.decode_posteriors(model)),
                        ("dump_1024", 1024, lambda: ctx.decode_posteriors(model))):
        use(n)
        line = {"config": "c1_" + name, "pairs": int(n), "groups": (int(n) + 63) // 64}
        for rep in range(2):
            for mode, key in ((0, "two_waves_ms"), (1, "one_wave_ms")):
                ctx.set_two_wave_windows(mode)
                line.setdefault(key, []).append(kernel_ms(fn, 3))
                line["waves_per_window_" + key[:3]] = ctx.last_waves_per_window()
        algo = float(n) * pm.S * (8 * pm.K + 0.25)
        line["frac_two_waves"] = algo / (min(line["two_waves_ms"]) / 1e3) / 8e12
        line["frac_one_wave"] = algo / (min(line["one_wave_ms"]) / 1e3) / 8e12
        print(json.dumps(line))
    ctx.close()


def k256(K=256):
    pm, bits, _, _ = prepared(600, 3000, K)  # 179 700 pairs = 2808 groups: every resident wave has work
    pairs = all_pairs(300)
    ctx = capi.Context(0)
    long_job_workspace(ctx)
    model = ctx.create_model(pm)
    ctx.upload_haps(bits, pm.S)
    pr = pairs.view(capi.PAIR_DTYPE).reshape(-1)
    ctx.upload_worklist(pr, capi.whole_sequence_groups(len(pr), pm.S, batch=64))
    dt, rec = timed(lambda: (ctx.decode_ibd_launch(model), ctx.decode_ibd_fetch())[1], reps=1)
    print(json.dumps({"config": f"k{K}_ibd", "pairs": len(pr), "sites": pm.S, "K": pm.K, "seconds": dt,
                      "pairs_per_s": len(pr) / dt, "pair_sites_per_s": len(pr) * pm.S / dt, "kernel_ms": ctx.last_kernel_ms(),
                      "algorithmic_GBps": len(pr) * pm.S * (8 * pm.K + 0.25) / dt / 1e9, "records": int(rec.size)}))
    ctx.close()


def seq(K=69):
    """Sequence mode (decodingSequence, scope row f4): two steps per site; 600 haplotypes x 3000 sites, all pairs."""
    tables = synth.make_model_tables(K)
    haps = synth.make_haps(600, 3000, seed=1234, cm_per_mb=1.2, bp_per_site=2500, switch_per_cm=2.0)
    data = api.Data.from_arrays(haps.alleles, haps.bp, haps.cm, True, True)
    dq = api.decoding_quantities_from_tables(tables)
    p = api.DecodingParams()
    p.FastSMC = True
    p.decodingModeString = "sequence"
    p.foldData = True
    p.usingCSFS = True
    p.batchSize = 32
    p.time = 50
    p.noConditionalAgeEstimates = True
    p.doPerPairPosteriorMean = True
    p.doPerPairMAP = True
    p.useKnownSeed = True
    p.hashing = False
    hmm = api.HMM(data, dq, p)
    pm = api.PreparedModelView(hmm.preparedModel())
    pairs = all_pairs(300)
    ctx = capi.Context(0)
    long_job_workspace(ctx)
    model = ctx.create_model(pm)
    ctx.upload_haps(data.packed_bits(), pm.S)
    pr = pairs.view(capi.PAIR_DTYPE).reshape(-1)
    ctx.upload_worklist(pr, capi.whole_sequence_groups(len(pr), pm.S, batch=64))
    dt, rec = timed(lambda: (ctx.decode_ibd_launch(model), ctx.decode_ibd_fetch())[1], reps=1)
    # two transition steps per site: the algorithmic bytes of the contract (one alpha row stored and read per site) stay
    print(json.dumps({"config": f"seq_k{K}_ibd", "pairs": len(pr), "sites": pm.S, "K": pm.K, "seconds": dt,
                      "kernel_member": ctx.last_kernel(), "pair_sites_per_s": len(pr) * pm.S / dt,
                      "kernel_ms": ctx.last_kernel_ms(),
                      "algorithmic_GBps": len(pr) * pm.S * (8 * pm.K + 0.25) / dt / 1e9, "records": int(rec.size)}))
    ctx.close()


def hashing():
    from oracle import oracle as O  # only for the table-key selection helper used by the tests as well

    n_hap, n_sites = 1000, 50000
    tables = synth.make_model_tables(69)
    haps = synth.make_haps(n_hap, n_sites, seed=1234)
    with tempfile.TemporaryDirectory() as d:
        root = os.path.join(d, "syn")
        t0 = time.perf_counter()
        synth.write_haps_files(root, haps)
        gen = (haps.cm / 100.0).astype(np.float32)
        used = np.unique(np.concatenate([[0.0], O.step_rows(tables.keys, gen)[1][1:]]))
        t = copy.copy(tables)
        sel = np.nonzero(np.isin(t.keys, used.astype(np.float32)))[0]
        t.keys, t.D, t.B, t.U, t.RR = t.keys[sel], t.D[sel], t.B[sel], t.U[sel], t.RR[sel]
        synth.write_decoding_quantities(root + ".decodingQuantities.gz", t)
        t_write = time.perf_counter() - t0
        p = api.DecodingParams()
        p.inFileRoot = root
        p.decodingQuantFile = root + ".decodingQuantities.gz"
        p.outFileRoot = os.path.join(d, "out")
        p.decodingModeString = "array"
        p.foldData = True
        p.usingCSFS = True
        p.batchSize = 32
        p.recallThreshold = 3
        p.min_m = 1.5
        p.hashing = True
        p.FastSMC = True
        p.BIN_OUT = False
        p.outputIbdSegmentLength = True
        p.time = 50
        p.noConditionalAgeEstimates = True
        p.doPerPairMAP = True
        p.doPerPairPosteriorMean = True
        p.useKnownSeed = True
        assert p.validateParamsFastSMC()
        t0 = time.perf_counter()
        f = api.FastSMC(p)
        t_init = time.perf_counter() - t0
        t0 = time.perf_counter()
        f.run()
        t_run = time.perf_counter() - t0
        n_lines = sum(1 for _ in gzip.open(p.outFileRoot + ".1.1.FastSMC.ibd.gz", "rt"))
    print(json.dumps({"config": "hashing_c2_files", "haplotypes": n_hap, "sites": n_sites, "K": 69,
                      "write_inputs_s": t_write, "construct_s": t_init, "run_s": t_run, "ibd_lines": n_lines,
                      "all_pairs": n_hap * (n_hap - 1) // 2 + 0}))



def run_c1_asmc(n_hap=300, n_sites=6760):
    """The reference's ONE published timing -- `ASMC_regression [HMM_regression]` (test_regression.cpp:23-68,
    time_regression.py:1-3: read the n300 array example, build the HMM, decodeAll with doPosteriorSums over all 44 850
    pairs; median 51.97 s) -- through the PRODUCT path on files of that shape: Data + HMM construction, then
    ASMC.decodeAllInJob() and the [sites][states] sums back in numpy."""
    from oracle import oracle as O

    tables = synth.make_model_tables(69)
    haps = synth.make_haps(n_hap, n_sites, seed=1234)
    with tempfile.TemporaryDirectory() as d:
        root = os.path.join(d, "syn")
        synth.write_haps_files(root, haps, fastsmc_map=False)
        # (ASMC mode reads the plink map and computes gen = stof(cM) / 100.f in float, Data.cpp:186: the keys of both)
        gen = (haps.cm / 100.0).astype(np.float32)
        gen32 = np.array([np.float32(np.float32(c) / np.float32(100.0)) for c in haps.cm], np.float32)
        used = np.unique(np.concatenate([[0.0], O.step_rows(tables.keys, gen)[1][1:], O.step_rows(tables.keys, gen32)[1][1:]]))
        t = copy.copy(tables)
        sel = np.nonzero(np.isin(t.keys, used.astype(np.float32)))[0]
        t.keys, t.D, t.B, t.U, t.RR = t.keys[sel], t.D[sel], t.B[sel], t.U[sel], t.RR[sel]
        synth.write_decoding_quantities(root + ".decodingQuantities.gz", t)
        # the constructor call of test_regression.cpp:27-41 (doPosteriorSums = true)
        p = api.DecodingParams(root, root + ".decodingQuantities.gz", "", 1, 1, "array", False, True, False, False, 0.0,
                               False, True)
        p.useKnownSeed = True
        for rep in range(2):  # (the first job of a process pays the HIP runtime's start-up)
            t0 = time.perf_counter()
            asmc = api.ASMC(p)
            t_init = time.perf_counter() - t0
            t0 = time.perf_counter()
            ret = asmc.decodeAllInJob()
            sums = np.asarray(ret.sumOverPairs)
            t_run = time.perf_counter() - t0
            n_pairs = n_hap * (n_hap - 1) // 2
            print(json.dumps({"config": "run_c1_asmc_published_job", "job_of_this_process": rep + 1, "haplotypes": n_hap,
                              "sites": n_sites, "K": 69, "pairs": n_pairs, "construct_s": t_init, "decode_all_s": t_run,
                              "job_s": t_init + t_run, "pairs_per_s_job": n_pairs / (t_init + t_run),
                              "reference_published_s": 51.97, "speedup_over_published": 51.97 / (t_init + t_run),
                              "sums_shape": list(sums.shape), "sums_checksum": float(np.float64(sums.sum()))}), flush=True)
            del asmc

def run_c2(n_hap=1000, n_sites=50000):
    """The PRODUCT path at the bench's size: FastSMC(params).run() end to end on C2 files -- read .hap.gz / .samples /
    .map / .decodingQuantities.gz, enumerate all pairs in batches of 32 (the reference default), decode, format and
    write the .ibd.gz -- next to the decode-only rate of bench.py."""
    import hashlib

    from oracle import oracle as O

    tables = synth.make_model_tables(69)
    haps = synth.make_haps(n_hap, n_sites, seed=1234)
    with tempfile.TemporaryDirectory() as d:
        root = os.path.join(d, "syn")
        t0 = time.perf_counter()
        synth.write_haps_files(root, haps)
        gen = (haps.cm / 100.0).astype(np.float32)
        used = np.unique(np.concatenate([[0.0], O.step_rows(tables.keys, gen)[1][1:]]))
        t = copy.copy(tables)
        sel = np.nonzero(np.isin(t.keys, used.astype(np.float32)))[0]
        t.keys, t.D, t.B, t.U, t.RR = t.keys[sel], t.D[sel], t.B[sel], t.U[sel], t.RR[sel]
        synth.write_decoding_quantities(root + ".decodingQuantities.gz", t)
        t_write = time.perf_counter() - t0
        p = api.DecodingParams()
        p.inFileRoot = root
        p.decodingQuantFile = root + ".decodingQuantities.gz"
        p.outFileRoot = os.path.join(d, "out")
        p.decodingModeString = "array"
        p.foldData = True
        p.usingCSFS = True
        p.batchSize = 32
        p.hashing = False
        p.FastSMC = True
        p.BIN_OUT = False
        p.outputIbdSegmentLength = True
        p.time = 50
        p.noConditionalAgeEstimates = True
        p.doPerPairMAP = True
        p.doPerPairPosteriorMean = True
        p.useKnownSeed = True
        assert p.validateParamsFastSMC()
        t0 = time.perf_counter()
        f = api.FastSMC(p)
        t_init = time.perf_counter() - t0
        t0 = time.perf_counter()
        f.run()
        t_run = time.perf_counter() - t0
        txt = gzip.open(p.outFileRoot + ".1.1.FastSMC.ibd.gz", "rb").read()
    n_pairs = n_hap * (n_hap - 1) // 2
    print(json.dumps({"config": "run_c2_product_path", "haplotypes": n_hap, "sites": n_sites, "K": 69,
                      "batchSize": 32, "pairs": n_pairs, "write_inputs_s": t_write, "construct_s": t_init,
                      "run_s": t_run, "pairs_per_s_run": n_pairs / t_run, "ibd_lines": txt.count(b"\n"),
                      "ibd_text_md5": hashlib.md5(txt).hexdigest()}))


def ingest_c3(n_hap=10000, n_sites=100000):
    """Host start-up at the C3 shape (10 000 haplotypes x 100 000 sites: 2 GB of haps text): Data(params) -- one
    multi-threaded pass over the .hap.gz -- and HMM(data, params) -- emission preparation, 3 x sites shuffles of cohort
    size -- timed separately, no GPU involved (the engine opens at the first decode)."""
    import zlib

    from oracle import oracle as O

    rng = np.random.default_rng(1234)
    tables = synth.make_model_tables(69)
    d = os.environ.get("FSMC_INGEST_DIR") or tempfile.mkdtemp()
    root = os.path.join(d, "c3")
    t0 = time.perf_counter()
    bp = np.cumsum(rng.integers(1, 600, n_sites)).astype(np.int64)
    cm = bp.astype(np.float64) * 1e-6
    freq = np.minimum(0.5, 0.01 / rng.uniform(0.01, 0.5, n_sites))  # a folded 1/x spectrum, MAF >= 1 %
    text_bytes = 0
    with open(root + ".hap.gz", "wb") as f:
        for s0 in range(0, n_sites, 500):  # one gzip member per 500 sites (a multi-member stream is a valid .gz)
            n = min(500, n_sites - s0)
            alleles = (rng.random((n, n_hap), dtype=np.float32) < freq[s0:s0 + n, None]).astype(np.uint8)
            body = np.empty((n, 2 * n_hap + 1), np.uint8)
            body[:, 0::2] = 32
            body[:, 1::2] = alleles + 48
            body[:, -1] = 10
            out = bytearray()
            for i in range(n):
                out += f"1:{int(bp[s0 + i])}_A_G SNP{s0 + i} {int(bp[s0 + i])} A G".encode()
                out += body[i].tobytes()
            text_bytes += len(out)
            co = zlib.compressobj(1, zlib.DEFLATED, 31)
            f.write(co.compress(bytes(out)) + co.flush())
    with open(root + ".samples", "w") as f:
        f.write("ID_1 ID_2 missing\n0 0 0\n")
        for i in range(n_hap // 2):
            f.write(f"1_{i + 1} 1_{i + 1} 0\n")
    with open(root + ".map", "w") as f:
        for s in range(n_sites):
            f.write(f"{int(bp[s])}\t1.0\t{float(cm[s])!r}\n")
    gen = (cm / 100.0).astype(np.float32)
    used = np.unique(np.concatenate([[0.0], O.step_rows(tables.keys, gen)[1][1:]]))
    t = copy.copy(tables)
    sel = np.nonzero(np.isin(t.keys, used.astype(np.float32)))[0]
    t.keys, t.D, t.B, t.U, t.RR = t.keys[sel], t.D[sel], t.B[sel], t.U[sel], t.RR[sel]
    synth.write_decoding_quantities(root + ".decodingQuantities.gz", t)
    t_write = time.perf_counter() - t0
    p = api.DecodingParams()
    p.inFileRoot = root
    p.decodingQuantFile = root + ".decodingQuantities.gz"
    p.outFileRoot = os.path.join(d, "out")
    p.decodingModeString = "array"
    p.foldData = True
    p.usingCSFS = True
    p.batchSize = 32
    p.hashing = False
    p.FastSMC = True
    p.time = 50
    p.useKnownSeed = True
    res = {"config": "ingest_c3", "haplotypes": n_hap, "sites": n_sites, "haps_text_bytes": text_bytes,
           "hap_gz_bytes": os.path.getsize(root + ".hap.gz"), "write_inputs_s": t_write,
           "host_threads": int(os.environ.get("FSMC_HOST_THREADS", "0")) or min(16, os.cpu_count() or 1)}
    t0 = time.perf_counter()
    data = api.Data(p)
    res["data_construct_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    hmm = api.HMM(data, p)
    res["hmm_construct_s"] = time.perf_counter() - t0
    res["construct_s"] = res["data_construct_s"] + res["hmm_construct_s"]
    res["sites_read"] = int(data.sites)
    res["haps_text_GB_per_s"] = text_bytes / 1e9 / res["data_construct_s"]
    del hmm
    print(json.dumps(res), flush=True)
    for ext in (".hap.gz", ".samples", ".map", ".decodingQuantities.gz"):
        os.remove(root + ext)


def c5_job(n_ind=8192, n_sites=20000, jobs=4, job=2):
    """One job of a biobank-scale run (BASELINE config 5): 16 384 haplotypes x 20 000 sites, hashing on, job 2 of 4 = the
    off-diagonal square of 67 M pairs: FastSMC(params) construction (one-pass reader + emission preparation) and run()
    (identification on the device, batching, paired windowed decode, binary output), timed."""
    from oracle import oracle as O

    haps = synth.make_haps_blocked(2 * n_ind, n_sites, seed=99, n_founders=48, cm_per_mb=1.0, switch_per_cm=0.3)
    tables = synth.make_model_tables(69)
    with tempfile.TemporaryDirectory() as d:
        root = os.path.join(d, "c5")
        synth.write_haps_files_fast(root, haps)
        gen = (haps.cm / np.float32(100.0)).astype(np.float32)
        used = np.unique(np.concatenate([[0.0], O.step_rows(tables.keys, gen)[1][1:]]))
        t = copy.copy(tables)
        sel = np.nonzero(np.isin(t.keys, used.astype(np.float32)))[0]
        t.keys, t.D, t.B, t.U, t.RR = t.keys[sel], t.D[sel], t.B[sel], t.U[sel], t.RR[sel]
        synth.write_decoding_quantities(root + ".decodingQuantities.gz", t)
        p = api.DecodingParams()
        p.inFileRoot = root
        p.decodingQuantFile = root + ".decodingQuantities.gz"
        p.outFileRoot = os.path.join(d, "out")
        p.decodingModeString = "array"
        p.foldData = True
        p.usingCSFS = True
        p.batchSize = 32
        p.min_m = 1.0
        p.hashing = True
        p.FastSMC = True
        p.BIN_OUT = True
        p.outputIbdSegmentLength = True
        p.time = 50
        p.noConditionalAgeEstimates = True
        p.doPerPairMAP = True
        p.doPerPairPosteriorMean = True
        p.useKnownSeed = True
        p.jobs, p.jobInd = jobs, job
        assert p.validateParamsFastSMC()
        t0 = time.perf_counter()
        f = api.FastSMC(p)
        t_init = time.perf_counter() - t0
        t0 = time.perf_counter()
        f.run()
        t_run = time.perf_counter() - t0
        segs = int(f.hmm().getNumSegmentsDetected())
        t0 = time.perf_counter()
        cands = api.hashingCandidatesDevice(api.Data(p), p)
        t_ident = time.perf_counter() - t0
        size = os.path.getsize(f.outputFileName())
    w = np.array([c[3] - c[2] for c in cands], np.int64)
    print(json.dumps({"config": "c5_job_window", "haplotypes": 2 * n_ind, "sites": n_sites, "jobs": jobs, "jobInd": job,
                      "pairs_in_window": (n_ind // 2) * (n_ind // 2) * 4, "candidates": len(cands),
                      "candidate_sites_median": float(np.median(w)), "candidate_pair_sites": int(w.sum()),
                      "segments": segs, "construct_s": t_init, "run_s": t_run,
                      "reread_plus_identify_s": t_ident, "bibd_bytes": size}), flush=True)


def short_windows():
    """The hashing regime (C5): batches of 32 pairs, each with its own short decode window (384-site median, as the
    C1 hashing run of SURVEY.md §0.9) -- pair-sites/s of the IBD decode alone."""
    pm, bits, _, _ = prepared(1000, 50000, 69)
    rng = np.random.default_rng(7)
    n_groups = 60000
    lens = rng.choice([320, 384, 384, 448, 640, 1024, 5504], size=n_groups, p=[0.2, 0.3, 0.2, 0.1, 0.1, 0.08, 0.02])
    starts = rng.integers(0, pm.S - lens)
    groups = np.zeros(n_groups, capi.GROUP_DTYPE)
    groups["first_pair"] = np.arange(n_groups) * 32
    groups["n_pairs"] = 32
    groups["from"] = starts
    groups["to"] = starts + lens
    groups["scan_from"] = starts + 16
    groups["scan_to"] = starts + lens - 16
    a = rng.integers(0, 1000, size=n_groups * 32).astype(np.uint32)
    b = (a + 1 + rng.integers(0, 998, size=a.size).astype(np.uint32)) % 1000
    pr = np.stack([a, b], axis=1).astype(np.uint32).view(capi.PAIR_DTYPE).reshape(-1)
    ctx = capi.Context(0)
    model = ctx.create_model(pm)
    ctx.upload_haps(bits, pm.S)
    ctx.upload_worklist(pr, groups)
    ps = float((lens.astype(np.int64) * 32).sum())
    # workspace: the limit a long job sets or has earned (the regime's steady state: every window whole, the long ones
    # pair up too), then pairing off, then -- in a fresh context -- the library's own policy for a job this short
    # (memory is earned by the work done: the few 5504-site windows are decoded in chunks)
    ctx.set_workspace_limit(int(0.8 * ctx.info()["hbm_bytes"]))
    for pairing, policy in ((1, "limit 0.8 of HBM"), (0, "limit 0.8 of HBM"), (1, "earned (default)")):
        if policy.startswith("earned"):
            ctx.close()
            ctx = capi.Context(0)
            model = ctx.create_model(pm)
            ctx.upload_haps(bits, pm.S)
            ctx.upload_worklist(pr, groups)
        ctx.set_pairing(pairing)
        dt, rec = timed(lambda: (ctx.decode_ibd_launch(model), ctx.decode_ibd_fetch())[1])
        print(json.dumps({"config": "short_windows_ibd", "pairing": pairing, "workspace": policy,
                          "wave_items": ctx.last_items(),
                          "beta_stride": ctx.last_beta_stride(), "groups": n_groups, "pairs": int(pr.size),
                          "pair_sites": ps, "seconds": dt, "kernel_ms": ctx.last_kernel_ms(),
                          "pair_sites_per_s": ps / dt, "pairs_per_s": pr.size / dt,
                          "algorithmic_GBps": ps * (8 * 69 + 0.25) / dt / 1e9, "records": int(rec.size),
                          "plan": ctx.info()}))
    ctx.close()


def identify():
    """fsmc_identify: every pair of the cohort against every complete word (passes 1 and 3 each do n^2/2 * words
    64-bit compares out of LDS).  LDS traffic per pair-word: 10 bytes (five 8-byte reads serve four pairs)."""
    for n_hap, S in ((1000, 50000), (10000, 100000)):
        haps = (synth.make_haps if n_hap <= 1000 else synth.make_haps_blocked)(n_hap, S, seed=1234)
        W = S // 64
        words = np.packbits(haps.alleles[:, :W * 64].reshape(n_hap, W, 64), axis=2,
                            bitorder="little").view(np.uint64).reshape(n_hap, W)
        gen = (haps.cm / 100.0).astype(np.float32)
        ids = np.arange(n_hap, dtype=np.uint32)
        ctx = capi.Context(0)
        rec = ctx.identify(words, ids, gen, min_m=1.0)  # warm-up (and the buffer size)
        t0 = time.perf_counter()
        rec = ctx.identify(words, ids, gen, min_m=1.0)
        dt = time.perf_counter() - t0
        ms = ctx.last_kernel_ms()
        # the other knobs of the reference (general match kernel; seeds split by the words read ahead)
        knobs = {}
        for name, kw in (("haploid_false", dict(haploid=False)), ("max_seeds_50", dict(max_seeds=50)),
                         ("word_size_32", dict(word_size=32))):
            w2 = words
            if "word_size" in kw:
                W2 = S // 32
                w2 = np.packbits(haps.alleles[:, :W2 * 32].reshape(n_hap, W2, 32), axis=2,
                                 bitorder="little").view(np.uint32).reshape(n_hap, W2).astype(np.uint64)
            ctx.identify(w2, ids, gen, min_m=1.0, **kw)
            t0 = time.perf_counter()
            r2 = ctx.identify(w2, ids, gen, min_m=1.0, **kw)
            knobs[name] = {"candidates": int(r2.size), "kernels_ms": ctx.last_kernel_ms(),
                           "call_s": time.perf_counter() - t0}
        ctx.close()
        pair_words = n_hap * (n_hap - 1) / 2 * W
        out = {"config": "identify", "haplotypes": n_hap, "sites": S, "words": W, "pairs": n_hap * (n_hap - 1) // 2,
               "candidates": int(rec.size), "kernels_ms": ms, "call_s": dt,
               "pair_words_per_s_two_passes": 2 * pair_words / (ms * 1e-3),
               "lds_GBps": 2 * pair_words * 10 / (ms * 1e-3) / 1e9, "lds_peak_GBps": 128 * 256 * 2.4,
               "other_knobs": knobs}
        if n_hap <= 1000:
            data = api.Data.from_arrays(haps.alleles, haps.bp, haps.cm, True, True)
            p = api.DecodingParams()
            p.FastSMC = True
            p.hashing = True
            p.min_m = 1.0
            t0 = time.perf_counter()
            host = api.hashingCandidates(data, p)
            out["host_restatement_s"] = time.perf_counter() - t0
            out["host_equal"] = [tuple(c) for c in host] == [(int(r["hap_a"]), int(r["hap_b"]), int(r["from"]), int(r["to"]))
                                                            for r in rec]
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["c1", "k256", "hashing"]
    for w in what:
        if w[0] == "k" and w[1:].isdigit():  # the 600 x 3000 list with a model of that many states
            k256(int(w[1:]))
            continue
        {"c1": c1, "c1_consumers": c1_consumers, "k256": k256, "k100": lambda: k256(100), "k128": lambda: k256(128), "k192": lambda: k256(192), "k300": lambda: k256(300), "k320": lambda: k256(320), "k350": lambda: k256(350), "k402": lambda: k256(402),
         "k448": lambda: k256(448), "k500": lambda: k256(500), "k600": lambda: k256(600), "hashing": hashing,
         "short": short_windows, "run_c2": run_c2, "run_c1_asmc": run_c1_asmc, "ingest_c3": ingest_c3, "c5_job": c5_job,
         "ingest_small": lambda: ingest_c3(2000, 20000), "identify": identify, "seq": seq, "seq100": lambda: seq(100)}[w]()
