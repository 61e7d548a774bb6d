"""The bench workload itself (BASELINE.json configs[1]: 1000 haplotypes x 50000 sites, K = 69, all 499500 pairs --
the chunked beta stream with checkpoints, 2048 resident waves, the dynamic group queue) checked where the oracle
can still follow and through size-independent properties elsewhere:
  * at least 2048 pairs sampled across the work list (2048 evenly spaced + 512 among those with segments): every field of
    every IBD record bit-identical to the oracle's (its -mavx2 build on every host core: ~5 s);
  * all records: ordered by (pair, start), inside the sequence, segments of a pair disjoint, score <= 1;
  * a second launch returns the identical record stream (no dependence on wave scheduling)."""
import numpy as np
import pytest

import bench
from fastsmc_amd import capi, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def test_c2_workload_sampled_parity_and_invariants():
    n_hap, S, K = 1000, 50000, 69
    pm, bits, haps, _ = bench.build_problem(n_hap, S, K, seed=1234)
    pairs = bench.all_pairs(n_hap // 2)
    n_pairs = pairs.shape[0]
    assert n_pairs == 499500
    ctx = capi.Context(0)
    try:
        model = ctx.create_model(pm)
        ctx.upload_haps(bits, pm.S)
        ctx.upload_worklist(pairs.view(capi.PAIR_DTYPE).reshape(-1), capi.whole_sequence_groups(n_pairs, pm.S, batch=64))
        flags = capi.FSMC_WANT_MEAN | capi.FSMC_WANT_MAP
        ctx.decode_ibd_launch(model, flags)
        rec = ctx.decode_ibd_fetch()
        ctx.decode_ibd_launch(model, flags)
        again = ctx.decode_ibd_fetch()
        info = ctx.info()
    finally:
        ctx.close()
    assert info["chunk_sites"] < S, "the full-size workload must exercise the checkpointed (chunked) path"
    assert rec.tobytes() == again.tobytes()

    # invariants over all records
    assert rec.size > 5000
    pair, start, end = rec["pair"].astype(np.int64), rec["start"].astype(np.int64), rec["end"].astype(np.int64)
    assert pair.min() >= 0 and pair.max() < n_pairs
    assert (start >= 0).all() and (end < S).all() and (start <= end).all()
    key = pair * S + start
    assert (np.diff(key) > 0).all()  # ordered by (pair, start), no duplicates
    same = pair[1:] == pair[:-1]
    assert (start[1:][same] > end[:-1][same]).all()  # segments of one pair do not overlap
    score = rec["prob"].astype(np.float64) / (end - start + 1)
    assert (score > 0).all() and (score <= 1.0 + 1e-5).all()
    assert np.isfinite(rec["post_mean"]).all() and (rec["post_mean"] > 0).all() and (rec["map"] > 0).all()

    # sampled pairs against the oracle: 2048 pairs x 50000 sites on every host core with the oracle's -mavx2 build
    # (tests/test_oracle_builds.py holds it bit-identical to the checker build) -- 2048 evenly spaced over the work
    # list + 512 evenly spaced over the pairs that have segments
    with_segments = np.unique(rec["pair"]).astype(np.int64)
    sample = np.unique(np.concatenate([np.linspace(0, n_pairs - 1, 2048).astype(np.int64),
                                       with_segments[np.linspace(0, with_segments.size - 1, 512).astype(np.int64)]]))
    assert sample.size >= 2048
    _, _, flipped = synth.fold_and_pack(haps.alleles)
    folded = np.where(flipped[None, :], 1 - haps.alleles, haps.alleles).astype(np.uint8)
    om = O.PreparedModel(K=pm.K, S=pm.S, pi=pm.pi, col_ratios=pm.col_ratios, exp_times=pm.exp_times, D=pm.D, B=pm.B,
                         U=pm.U, RR=pm.RR, step_row=pm.step_row, e1=pm.e1, e0m1=pm.e0m1, e2m0=pm.e2m0,
                         gen=np.zeros(pm.S, np.float32), phys=np.zeros(pm.S, np.int32),
                         state_threshold=int(pm.state_threshold), age_threshold=int(pm.age_threshold),
                         probability_threshold=np.float32(pm.probability_threshold))
    import os

    O.select_build("avx2")
    try:
        want = O.decode_pairs_ibd(om, folded, [tuple(int(x) for x in pairs[i]) for i in sample], batch_size=32,
                                  threads=max(1, min(len(os.sched_getaffinity(0)), 16)))
    finally:
        O.select_build("ref")
    got = rec[np.isin(rec["pair"], sample)]
    assert got.size == want.size and want.size > 500
    np.testing.assert_array_equal(got["pair"], sample[want["pair"]])
    for f_got, f_want in (("start", "start"), ("end", "end"), ("prob", "prob"), ("post_mean", "postMean"),
                          ("map", "map")):
        np.testing.assert_array_equal(got[f_got], want[f_want], err_msg=f_got)
