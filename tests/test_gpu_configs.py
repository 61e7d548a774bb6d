"""Every BASELINE.json configuration on the GPU, checked where the oracle can still follow (sampled pairs, every field
of every record bit-identical) and through size-independent properties elsewhere (order, range, disjointness,
determinism).  C2 has its own file (test_gpu_full_size.py).

  C1  the reference's FASTSMC_EXAMPLE data (300 haplotypes x 6760 sites; tests/golden/fastsmc_example.*), job 7 of 9
      of the no-hashing regression shape (test_fastsmc_regression.cpp:97-161), synthetic map + 69-state model:
      FastSMC.run() text byte-identical to oracle + record formatter.
  C3  the per-GPU shard shape of the 10 000 x 100 000 cohort: windows of 100 000 sites, K = 69 (49 chunks); 512
      sampled pairs against the oracle.
  C4  K = 256, windows of 200 000 sites through the chunked wave-group kernel (four waves per group, lane = pair);
      128 sampled pairs against the oracle.
  C5  the hashing regime: 10 240 batches of 32 pairs, each with its own window of 320 ... 5504 sites.
"""
import copy
import gzip
import os
import shutil

import numpy as np
import pytest

import bench
from fastsmc_amd import api, capi, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIELDS = (("start", "start"), ("end", "end"), ("prob", "prob"), ("post_mean", "postMean"), ("map", "map"))


def _oracle_model(pm):
    return O.PreparedModel(K=pm.K, S=pm.S, pi=pm.pi, col_ratios=pm.col_ratios, exp_times=pm.exp_times, D=pm.D, B=pm.B,
                           U=pm.U, RR=pm.RR, step_row=pm.step_row, e1=pm.e1, e0m1=pm.e0m1, e2m0=pm.e2m0,
                           gen=np.zeros(pm.S, np.float32), phys=np.zeros(pm.S, np.int32),
                           state_threshold=int(pm.state_threshold), age_threshold=int(pm.age_threshold),
                           probability_threshold=np.float32(pm.probability_threshold))


def _invariants(rec, n_pairs, S):
    pair, start, end = rec["pair"].astype(np.int64), rec["start"].astype(np.int64), rec["end"].astype(np.int64)
    assert pair.min() >= 0 and pair.max() < n_pairs
    assert (start >= 0).all() and (end < S).all() and (start <= end).all()
    assert (np.diff(pair * S + start) > 0).all()  # ordered by (pair, start), no duplicates
    same = pair[1:] == pair[:-1]
    assert (start[1:][same] > end[:-1][same]).all()  # segments of one pair do not overlap
    score = rec["prob"].astype(np.float64) / (end - start + 1)
    assert (score > 0).all() and (score <= 1.0 + 1e-5).all()


def _whole_sequence_case(n_hap, S, K, n_pairs, n_sample, want_member, time=50, seed=1234, oracle_batch=32,
                         oracle_threads=None, chunk=2048, group_mb=None, want_resident=None):
    """All-pairs-style list over whole-sequence windows: decode twice, check invariants, sample against the oracle (its
    -mavx2 build, which tests/test_oracle_builds.py holds bit-identical to the checker build, one batch per host
    thread; `oracle_batch` bounds a thread's two S x K x batch buffers -- per-pair results do not depend on it)."""
    tables = synth.make_model_tables(K)
    haps = synth.make_haps(n_hap, S, seed=seed)
    data = api.Data.from_arrays(haps.alleles, haps.bp, haps.cm, True, True)
    p = bench.fastsmc_params()
    p.time = time
    hmm = api.HMM(data, api.decoding_quantities_from_tables(tables), p)
    pm = api.PreparedModelView(hmm.preparedModel())
    pairs = bench.all_pairs(n_hap // 2)[:n_pairs]
    ctx = capi.Context(0)
    try:
        model = ctx.create_model(pm)
        ctx.upload_haps(data.packed_bits(), pm.S)
        ctx.upload_worklist(pairs.view(capi.PAIR_DTYPE).reshape(-1), capi.whole_sequence_groups(n_pairs, pm.S, batch=64))
        # the chunk length the full-size list gets (2048 resident waves share the workspace); with the few waves
        # of this sample every beta row would fit and the rebuild pass would never run
        ctx.set_chunk_sites(chunk)
        # 40 MB a wave, as at full size (`group_mb`: a case that wants room for resident chunks, or has wider rows)
        ctx.set_workspace_limit((n_pairs // 64) * ((group_mb << 20) if group_mb else (4 if K > 80 else 1) * (40 << 20)))
        ctx.decode_ibd_launch(model)
        rec = ctx.decode_ibd_fetch()
        ctx.decode_ibd_launch(model)
        again = ctx.decode_ibd_fetch()
        info, member, resident = ctx.info(), ctx.last_kernel(), ctx.last_resident_chunks()
    finally:
        ctx.close()
    assert member == want_member
    assert info["max_chunks"] > 1, "the configuration must run through the checkpointed (chunked) layout"
    if want_resident is not None:
        assert resident == want_resident
    assert rec.tobytes() == again.tobytes()
    assert rec.size > 20
    _invariants(rec, n_pairs, S)
    with_segments = np.unique(rec["pair"]).astype(np.int64)
    # n_sample pairs evenly spaced over the list, and a quarter as many again among the pairs that have segments
    sample = np.unique(np.concatenate([np.linspace(0, n_pairs - 1, n_sample).astype(np.int64),
                                       with_segments[np.linspace(0, with_segments.size - 1,
                                                                 max(4, n_sample // 4)).astype(np.int64)]]))
    _, _, flipped = synth.fold_and_pack(haps.alleles)
    folded = np.where(flipped[None, :], 1 - haps.alleles, haps.alleles).astype(np.uint8)
    threads = oracle_threads or max(1, min(len(os.sched_getaffinity(0)), 16))
    O.select_build("avx2")
    try:
        want = O.decode_pairs_ibd(_oracle_model(pm), folded, [tuple(int(x) for x in pairs[i]) for i in sample],
                                  batch_size=oracle_batch, threads=threads)
    finally:
        O.select_build("ref")
    assert sample.size >= n_sample
    got = rec[np.isin(rec["pair"], sample)]
    assert got.size == want.size and want.size > 0
    np.testing.assert_array_equal(got["pair"], sample[want["pair"]])
    for f_got, f_want in FIELDS:
        np.testing.assert_array_equal(got[f_got], want[f_want], err_msg=f_got)


def test_c3_shard_shape_100k_sites():
    # 512 of the 4096 pairs against the oracle (16 pairs a batch: 14 GB of oracle buffers on 16 threads)
    _whole_sequence_case(n_hap=128, S=100000, K=69, n_pairs=4096, n_sample=512, want_member=69, oracle_batch=16)


def test_c4_256_states_200k_sites_chunked_wide_model_kernel():
    # 128 of the 2048 pairs against the oracle (8 pairs a batch on 8 threads: 26 GB of oracle buffers, ~15 s)
    _whole_sequence_case(n_hap=128, S=200000, K=256, n_pairs=2048, n_sample=128, want_member=1064, time=200,
                         oracle_batch=8, oracle_threads=8)


def test_256_states_60k_sites_with_resident_chunks():
    # config 4's kernel with room for two resident chunks a group (what the full-size run gets from 80 % of
    # the card): 64 of the 1024 pairs against the oracle
    _whole_sequence_case(n_hap=128, S=60000, K=256, n_pairs=1024, n_sample=64, want_member=1064, time=200,
                         oracle_batch=8, oracle_threads=8, group_mb=400, want_resident=2)


def test_600_states_30k_sites_chunked_eight_wave_member_without_landing_zones():
    # a model beyond 512 states on long windows (rows of 160 KiB; chunks of 1024 sites): 64 of the 512 pairs
    _whole_sequence_case(n_hap=64, S=30000, K=600, n_pairs=512, n_sample=64, want_member=8080, time=200,
                         oracle_batch=8, oracle_threads=8, chunk=1024, group_mb=320, want_resident=0)


def test_c1_reference_example_data_job_7_of_9(tmp_path):
    # inputs: the reference's haplotypes and samples, a synthetic 1 cM/Mb map, the synthetic 69-state model
    root = str(tmp_path / "example")
    shutil.copy(os.path.join(GOLD, "fastsmc_example.hap.gz"), root + ".hap.gz")
    shutil.copy(os.path.join(GOLD, "fastsmc_example.samples"), root + ".samples")
    bp, rows = [], []
    for line in gzip.open(root + ".hap.gz", "rt"):
        t = line.split()
        bp.append(int(t[2]))
        rows.append(np.array(t[5:], dtype=np.uint8))
    alleles, bp = np.stack(rows, axis=1), np.array(bp, np.int64)
    cm = bp.astype(np.float64) * 1e-6
    with open(root + ".map", "w") as f:
        for x, c in zip(bp, cm):
            f.write(f"{int(x)}\t1.0\t{float(c)!r}\n")
    gen = (cm / 100.0).astype(np.float32)
    tables = synth.make_model_tables(69)
    used = np.unique(np.concatenate([[0.0], O.step_rows(tables.keys, gen)[1][1:]]))
    t = copy.copy(tables)
    sel = np.nonzero(np.isin(t.keys, used.astype(np.float32)))[0]
    t.keys, t.D, t.B, t.U, t.RR = t.keys[sel], t.D[sel], t.B[sel], t.U[sel], t.RR[sel]
    synth.write_decoding_quantities(root + ".decodingQuantities.gz", t)
    # the run of test_fastsmc_regression.cpp:97-122
    p = api.DecodingParams()
    p.inFileRoot, p.decodingQuantFile, p.outFileRoot = root, root + ".decodingQuantities.gz", str(tmp_path / "res")
    p.decodingModeString = "array"
    p.foldData = p.usingCSFS = True
    p.batchSize, p.recallThreshold, p.min_m = 32, 3, 1.5
    p.hashing, p.FastSMC, p.BIN_OUT = False, True, False
    p.outputIbdSegmentLength = True
    p.time = 50
    p.noConditionalAgeEstimates = p.doPerPairMAP = p.doPerPairPosteriorMean = True
    p.jobInd, p.jobs = 7, 9
    p.useKnownSeed = True
    assert p.validateParamsFastSMC()
    api.FastSMC(p).run()
    got = gzip.open(p.outFileRoot + ".7.9.FastSMC.ibd.gz", "rt").read()
    # what the reference's NO_SSE path writes for the same inputs: oracle + record formatter
    individuals = O.job_individuals(150, 9, 7)
    hrows = np.array([2 * d + h for d in individuals for h in (0, 1)])
    _, derived, flipped = synth.fold_and_pack(alleles)
    folded = np.where(flipped[None, :], 1 - alleles, alleles).astype(np.uint8)[hrows]
    pm = O.prepare_model(tables, gen, bp, derived, alleles.shape[0], time=50)
    pairs = O.enumerate_all_pairs(len(individuals), 9, 7)
    assert len(pairs) == 2211  # C1, job 7/9
    recs = O.decode_pairs_ibd(pm, folded, pairs, batch_size=32)
    ids = [f"1_{d + 1}" for d in individuals]
    want = O.format_ibd_text(recs, pairs, ids, ids, 1, bp, gen)
    assert want.count("\n") > 500
    assert got == want


def test_c5_hashing_regime_ten_thousand_windowed_batches():
    """10 240 batches of 32 pairs, windows of 320 ... 5504 sites (the distribution of the C1 hashing run, SURVEY.md
    §0.9) anywhere on a 50 000-site sequence, scan windows inside them: invariants + determinism over all batches,
    48 batches against the oracle bit for bit."""
    n_hap, S, K = 1000, 50000, 69
    pm, bits, haps, _ = bench.build_problem(n_hap, S, K, seed=1234)
    rng = np.random.default_rng(7)
    n_groups = 10240
    lens = rng.choice([320, 384, 384, 448, 640, 1024, 5504], size=n_groups, p=[0.2, 0.3, 0.2, 0.1, 0.1, 0.08, 0.02])
    starts = rng.integers(0, pm.S - lens)
    groups = np.zeros(n_groups, capi.GROUP_DTYPE)
    groups["first_pair"] = np.arange(n_groups) * 32
    groups["n_pairs"] = 32
    groups["from"], groups["to"] = starts, starts + lens
    groups["scan_from"], groups["scan_to"] = starts + 16, starts + lens - 16
    a = rng.integers(0, n_hap, size=n_groups * 32).astype(np.uint32)
    b = (a + 1 + rng.integers(0, n_hap - 2, size=a.size).astype(np.uint32)) % n_hap
    pairs = np.stack([a, b], axis=1).astype(np.uint32)
    ctx = capi.Context(0)
    try:
        model = ctx.create_model(pm)
        ctx.upload_haps(bits, pm.S)
        ctx.upload_worklist(pairs.view(capi.PAIR_DTYPE).reshape(-1), groups)
        # a short job under the library's own workspace policy: memory is earned by the work done (hipMalloc costs
        # 40 ms per GB), so the few 5504-site windows are decoded in chunks ...
        ctx.decode_ibd_launch(model)
        earned = ctx.decode_ibd_fetch()
        earned_chunks = ctx.info()["max_chunks"]
        # ... and with the limit a long job sets (or has earned) every window fits its wave's workspace whole and the
        # long ones pair up too: no rebuild pass.  The records do not depend on the plan.
        ctx.set_workspace_limit(int(0.5 * ctx.info()["hbm_bytes"]))
        ctx.decode_ibd_launch(model)
        rec = ctx.decode_ibd_fetch()
        ctx.decode_ibd_launch(model)
        again = ctx.decode_ibd_fetch()
        info = ctx.info()
    finally:
        ctx.close()
    assert rec.tobytes() == again.tobytes() and rec.size > 100
    assert rec.tobytes() == earned.tobytes()
    assert earned_chunks > 1 and info["max_chunks"] == 1
    _invariants(rec, pairs.shape[0], S)
    g_of = rec["pair"] // 32
    assert (rec["start"] >= groups["scan_from"][g_of]).all() and (rec["end"] < groups["scan_to"][g_of]).all()
    # sampled batches against the oracle (batch window decode + scan window, HMM.cpp:1199-1206)
    _, _, flipped = synth.fold_and_pack(haps.alleles)
    folded = np.where(flipped[None, :], 1 - haps.alleles, haps.alleles).astype(np.uint8)
    om = _oracle_model(pm)
    hit = np.unique(g_of)
    sample = np.unique(np.concatenate([np.linspace(0, n_groups - 1, 24).astype(np.int64),
                                       hit[np.linspace(0, hit.size - 1, 24).astype(np.int64)]]))
    for g in sample:
        frm, to = int(groups["from"][g]), int(groups["to"][g])
        sub = pairs[32 * g:32 * g + 32]
        ob = np.stack([(folded[x] ^ folded[y])[frm:to] for x, y in sub])
        hb = np.stack([(folded[x] & folded[y])[frm:to] for x, y in sub])
        post, _ = O.decode_batch(om, ob, hb, frm, to)
        full = np.zeros((S, pm.K, 32), np.float32)
        full[frm:to] = post[frm:to]
        want = np.concatenate([O.ibd_scan_pair(om, full, v, int(groups["scan_from"][g]), int(groups["scan_to"][g]),
                                               pair_ordinal=32 * int(g) + v) for v in range(32)])
        got = rec[g_of == g]
        assert got.size == want.size
        np.testing.assert_array_equal(got["pair"], want["pair"])
        for f_got, f_want in FIELDS:
            np.testing.assert_array_equal(got[f_got], want[f_want], err_msg=f"{f_got} batch {g}")
