"""BASELINE config 5 at the scale of ONE JOB of a biobank run: 16 384 haplotypes x 20 000 sites, hashing on, jobs a
perfect square so that the job's window is a real off-diagonal square (Data.cpp:62-80: job 2 of 4 = every pair between
the first and the second half of the individuals).  One FastSMC(params).run(): the one-pass reader, the identification
step on the device over 67 M pairs of the window (fsmc_identify), batching of the candidates in emission order, paired
windowed decode, record output.  Checked: size-independent properties of every record, 32 sampled batches against the
oracle bit for bit (the batch windows of HMM.cpp:555-636, 1199-1206), and determinism of a second run."""
import copy
import os

import numpy as np
import pytest

from fastsmc_amd import api, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu

N_IND, SITES, JOBS, JOB = 8192, 20000, 4, 2


@pytest.fixture(scope="module")
def cohort(tmp_path_factory):
    haps = synth.make_haps_blocked(2 * N_IND, SITES, seed=99, n_founders=48, cm_per_mb=1.0, switch_per_cm=0.3)
    tables = synth.make_model_tables(69)
    root = str(tmp_path_factory.mktemp("c5") / "c5")
    synth.write_haps_files_fast(root, haps)
    gen = (haps.cm / np.float32(100.0)).astype(np.float32)
    used = np.unique(np.concatenate([[0.0], O.step_rows(tables.keys, gen)[1][1:]]))
    t = copy.copy(tables)
    sel = np.nonzero(np.isin(t.keys, used.astype(np.float32)))[0]
    t.keys, t.D, t.B, t.U, t.RR = t.keys[sel], t.D[sel], t.B[sel], t.U[sel], t.RR[sel]
    synth.write_decoding_quantities(root + ".decodingQuantities.gz", t)
    return root, haps, tables, gen


def _params(root, out):
    p = api.DecodingParams()
    p.inFileRoot = root
    p.decodingQuantFile = root + ".decodingQuantities.gz"
    p.outFileRoot = out
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
    p.jobs, p.jobInd = JOBS, JOB
    assert p.validateParamsFastSMC()
    return p


def _run(root, out):
    f = api.FastSMC(_params(root, out))
    f.hmm().setKeepIbdRecords(True)
    f.run()
    return f.hmm().getIbdRecordArrays(), os.path.getsize(f.outputFileName())


def test_one_job_window_of_a_biobank_run(cohort, tmp_path):
    root, haps, tables, gen = cohort
    rec, size = _run(root, str(tmp_path / "a"))
    n = rec["pair"].size
    assert n > 1000 and size > 0
    # ---- properties of every record
    ia, ib = rec["hap_a"] // 2, rec["hap_b"] // 2
    lo, hi = np.minimum(ia, ib), np.maximum(ia, ib)
    assert (lo < N_IND // 2).all() and (hi >= N_IND // 2).all()  # the off-diagonal square: one individual from each window
    assert (rec["start"] >= 0).all() and (rec["end"] < SITES).all() and (rec["start"] <= rec["end"]).all()
    score = rec["prob"].astype(np.float64) / (rec["end"] - rec["start"] + 1)
    assert (score > 0).all() and (score <= 1.0 + 1e-6).all()
    assert np.isfinite(rec["post_mean"]).all() and (rec["map"] > 0).all()
    assert (np.diff(rec["pair"].astype(np.int64)) >= 0).all()  # candidates come out in their emission order
    same = np.diff(rec["pair"].astype(np.int64)) == 0
    assert (rec["start"][1:][same] > rec["end"][:-1][same]).all()  # segments of one candidate: ordered, disjoint

    # ---- the candidates the run decoded, in its emission order; 32 sampled batches against the oracle
    p = _params(root, str(tmp_path / "cands"))
    data = api.Data(p)
    cands = api.hashingCandidatesDevice(data, p)
    assert len(cands) > 10000 and int(rec["pair"].max()) < len(cands)
    _, derived, flipped = synth.fold_and_pack(haps.alleles)
    folded = np.where(flipped[None, :], 1 - haps.alleles, haps.alleles).astype(np.uint8)
    pm = O.prepare_model(tables, gen, haps.bp, derived, haps.alleles.shape[0], time=50)
    n_batches = (len(cands) + 31) // 32
    hit = np.unique(rec["pair"] // 32)
    sample = np.unique(np.concatenate([np.linspace(0, n_batches - 1, 12).astype(np.int64),
                                       hit[np.linspace(0, hit.size - 1, 20).astype(np.int64)]]))
    checked = 0
    for b in sample:
        batch = cands[32 * int(b):32 * int(b) + 32]
        start, end = min(c[2] for c in batch), max(c[3] for c in batch)
        frm, to = O.get_from_position(gen, start), O.get_to_position(gen, end)
        ob = np.stack([(folded[x] ^ folded[y])[frm:to] for x, y, _, _ in batch])
        hb = np.stack([(folded[x] & folded[y])[frm:to] for x, y, _, _ in batch])
        while ob.shape[0] % 4:
            ob = np.concatenate([ob, ob[-1:]])
            hb = np.concatenate([hb, hb[-1:]])
        post, _ = O.decode_batch(pm, ob, hb, frm, to)
        want = np.concatenate([O.ibd_scan_pair(pm, post, v, start, end, pair_ordinal=32 * int(b) + v)
                               for v in range(len(batch))])
        sel = (rec["pair"] // 32) == b
        assert int(sel.sum()) == want.size, f"batch {b}"
        np.testing.assert_array_equal(rec["pair"][sel], want["pair"])
        for got_f, want_f in (("start", "start"), ("end", "end"), ("prob", "prob"), ("post_mean", "postMean"),
                              ("map", "map")):
            np.testing.assert_array_equal(rec[got_f][sel], want[want_f], err_msg=f"{got_f} batch {b}")
        for k in np.nonzero(sel)[0][:3]:
            assert (int(rec["hap_a"][k]), int(rec["hap_b"][k])) == tuple(batch[int(rec["pair"][k]) - 32 * int(b)][:2])
        checked += want.size
    assert checked > 20

    # ---- a second run: the same records, the same file size
    again, size2 = _run(root, str(tmp_path / "b"))
    assert size2 == size
    for k in rec:
        np.testing.assert_array_equal(again[k], rec[k])
