"""End-to-end through the product's host API (pybind11 classes over the C ABI) on a real GPU:
FastSMC-mode runs must write byte-identical IBD text to what the oracle + the reference's record
format give, including the job-slicing of the pinned no-hashing regression shape (job 7 of 9)."""
import copy
import gzip

import numpy as np
import pytest

from fastsmc_amd import api, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def make_files(dirpath, sp=None):
    """The small problem as the reference's input files (.hap.gz / .samples / .map / .decodingQuantities.gz) under
    `dirpath`; returns the file root.  (Also called from child processes of other tests.)"""
    import os

    if sp is None:
        from conftest import build_small_problem

        sp = build_small_problem()
    root = os.path.join(str(dirpath), "syn")
    synth.write_haps_files(root, sp["haps"])
    used = np.unique(np.concatenate([[0.0], O.step_rows(sp["tables"].keys, sp["gen"])[1][1:]]))
    t = copy.copy(sp["tables"])
    sel = np.nonzero(np.isin(t.keys, used.astype(np.float32)))[0]
    t.keys, t.D, t.B, t.U, t.RR = t.keys[sel], t.D[sel], t.B[sel], t.U[sel], t.RR[sel]
    synth.write_decoding_quantities(root + ".decodingQuantities.gz", t)
    return root


@pytest.fixture(scope="module")
def files(small_problem, tmp_path_factory):
    return make_files(tmp_path_factory.mktemp("data"), small_problem)


def _params(root, out, **kw):
    p = api.DecodingParams()
    p.inFileRoot = root
    p.decodingQuantFile = root + ".decodingQuantities.gz"
    p.outFileRoot = out
    p.decodingModeString = "array"
    p.foldData = True
    p.usingCSFS = True
    p.batchSize = 32
    p.recallThreshold = 3
    p.min_m = 1.5
    p.hashing = False
    p.FastSMC = True
    p.BIN_OUT = False
    p.outputIbdSegmentLength = True
    p.time = 50
    p.noConditionalAgeEstimates = True
    p.doPerPairMAP = True
    p.doPerPairPosteriorMean = True
    p.useKnownSeed = True
    for k, v in kw.items():
        setattr(p, k, v)
    assert p.validateParamsFastSMC()
    return p


def _oracle_text(sp, individuals, jobs, job_ind):
    """What the reference would write: subset of individuals (job windows), pair slice, batches of 32."""
    haps = sp["haps"]
    rows = np.array([2 * d + h for d in individuals for h in (0, 1)])
    alleles = haps.alleles  # folding uses ALL samples of the file (Data.cpp:465-471)
    _, derived, flipped = synth.fold_and_pack(alleles)
    folded_all = np.where(flipped[None, :], 1 - alleles, alleles).astype(np.uint8)
    folded = folded_all[rows]
    pm = O.prepare_model(sp["tables"], sp["gen"], haps.bp, derived, alleles.shape[0], time=50)
    pairs = O.enumerate_all_pairs(len(individuals), jobs, job_ind)
    recs = O.decode_pairs_ibd(pm, folded, pairs, batch_size=32)
    ids = [f"1_{d + 1}" for d in individuals]
    return O.format_ibd_text(recs, pairs, ids, ids, 1, haps.bp, sp["gen"]), len(pairs)


def test_fastsmc_run_matches_oracle_text(files, small_problem, tmp_path):
    out = str(tmp_path / "res")
    p = _params(files, out)
    api.FastSMC(p).run()
    got = gzip.open(out + ".1.1.FastSMC.ibd.gz", "rt").read()
    want, n_pairs = _oracle_text(small_problem, list(range(32)), 1, 1)
    assert n_pairs == 2 * 32 * 32 - 32
    assert want.count("\n") > 100
    assert got == want


def test_fastsmc_job_7_of_9_matches_oracle_text(files, small_problem, tmp_path):
    """Shape of the reference's no-hashing regression (test_fastsmc_regression.cpp:97-161): jobInd 7 of 9 loads a
    subset of individuals (Data.cpp:62-80) and decodes a slice of that subset's pairs (HMM.cpp:319-321)."""
    out = str(tmp_path / "res79")
    p = _params(files, out, jobs=9, jobInd=7)
    f = api.FastSMC(p)
    f.run()
    got = gzip.open(out + ".7.9.FastSMC.ibd.gz", "rt").read()
    individuals = O.job_individuals(32, 9, 7)
    want, _ = _oracle_text(small_problem, individuals, 9, 7)
    assert got == want and want


def test_hmm_decode_single_pair_matches_oracle_posterior(files, small_problem, tmp_path):
    p = _params(files, str(tmp_path / "x"))
    data = api.Data(p)
    hmm = api.HMM(data, p)
    obs = hmm.makePairObs(1, 0, 2, 5)  # haplotype rows 0 and 11
    post = np.array(hmm.decode(obs), np.float32)  # [K][S]
    sp = small_problem
    ob = (sp["folded"][0] ^ sp["folded"][11])[None, :]
    hb = (sp["folded"][0] & sp["folded"][11])[None, :]
    want, _ = O.decode_batch(sp["model"], ob, hb, 0, sp["model"].S)
    np.testing.assert_array_equal(post, want[:, :, 0].T)
    assert np.array_equal(np.array(obs.obsBits, np.uint8), ob[0])


def test_decode_from_hashing_batches_windows_like_the_reference(files, small_problem, tmp_path):
    """Candidates pushed through HMM.decodeFromHashing: batch windows = union of the batch's candidate windows
    padded by 0.5 cM; every pair scanned over the un-padded union (HMM.cpp:470-502, 555-636, 1199-1206)."""
    out = str(tmp_path / "hash")
    p = _params(files, out, hashing=True)
    data = api.Data(p)
    hmm = api.HMM(data, p)
    hmm.setKeepIbdRecords(True)
    hmm.decodeAll(1, 1)
    rng = np.random.default_rng(5)
    S = small_problem["model"].S
    cands = []
    for _ in range(75):  # 2 full batches of 32 + a ragged one of 11
        a, b = sorted(rng.choice(64, 2, replace=False))
        f = int(rng.integers(0, S - 120))
        t = f + int(rng.integers(40, 110))
        cands.append((int(a), int(b), f, t))
        hmm.decodeFromHashing(int(a), int(b), f, t)
    hmm.finishFromHashing()
    got = hmm.getIbdRecords()
    recs, _ = _oracle_hashing_records(small_problem, small_problem["folded"], small_problem["model"], cands)
    want = [(cands[int(r["pair"])][0], cands[int(r["pair"])][1], int(r["start"]), int(r["end"]), float(r["prob"]),
             float(r["postMean"]), float(r["map"])) for r in recs]
    assert len(want) > 10
    assert [tuple(x) for x in got] == want


def _oracle_hashing_records(sp, folded, pm, cands):
    """Batches of 32 candidates in arrival order; window = union of the batch's candidate windows padded by 0.5 cM;
    every pair scanned over the un-padded union (HMM.cpp:470-502, 555-636, 1199-1206)."""
    gen = sp["gen"]
    recs = []
    for b0 in range(0, len(cands), 32):
        batch = cands[b0:b0 + 32]
        start = min(c[2] for c in batch)
        end = max(c[3] for c in batch)
        frm, to = O.get_from_position(gen, start), O.get_to_position(gen, end)
        ob = np.stack([(folded[a] ^ folded[b])[frm:to] for a, b, _, _ in batch])
        hb = np.stack([(folded[a] & folded[b])[frm:to] for a, b, _, _ in batch])
        while ob.shape[0] % 4:
            ob = np.concatenate([ob, ob[-1:]])
            hb = np.concatenate([hb, hb[-1:]])
        post, _ = O.decode_batch(pm, ob, hb, frm, to)
        for v in range(len(batch)):
            recs.append(O.ibd_scan_pair(pm, post, v, start, end, pair_ordinal=b0 + v))
    recs = np.concatenate(recs) if recs else np.zeros(0, O.IBD_DTYPE)
    return recs, [(c[0], c[1]) for c in cands]


@pytest.mark.parametrize("jobs,job_ind", [(1, 1), (4, 3)])
def test_fastsmc_run_with_hashing_matches_oracle_text(files, small_problem, tmp_path, jobs, job_ind):
    """FastSMC.run() with the identification step on (FastSMC.cpp:118-235): hashing candidates -> batched window
    decodes -> IBD text, against the restated pre-filter (tests/test_hashing.py) + the oracle's batches."""
    from test_hashing import restate_candidates

    sp = small_problem
    out = str(tmp_path / "hashrun")
    p = _params(files, out, hashing=True, min_m=1.0, jobs=jobs, jobInd=job_ind)
    api.FastSMC(p).run()
    got = gzip.open(out + f".{job_ind}.{jobs}.FastSMC.ibd.gz", "rt").read()
    individuals = O.job_individuals(32, jobs, job_ind) if jobs > 1 else list(range(32))
    alleles = sp["haps"].alleles
    cands = restate_candidates(alleles, sp["gen"], individuals, jobs=jobs, job_ind=job_ind, min_m=1.0)
    assert len(cands) >= 20
    rows = np.array([2 * d + h for d in individuals for h in (0, 1)])
    recs, pairs = _oracle_hashing_records(sp, sp["folded"][rows], sp["model"], cands)
    ids = [f"1_{d + 1}" for d in individuals]
    want = O.format_ibd_text(recs, pairs, ids, ids, 1, sp["haps"].bp, sp["gen"])
    assert want.count("\n") > 10
    assert got == want


@pytest.mark.parametrize("opts", [dict(haploid=False, min_m=0.4),
                                  dict(max_seeds=2, hashingWordSize=32, constReadAhead=6, min_m=0.8)])
def test_fastsmc_run_with_the_other_hashing_knobs(files, small_problem, tmp_path, opts):
    """The same run with matches keyed by individual pairs (haploid = false: the candidates name haplotype 1 of each
    individual and include an individual with itself, HMM.cpp:483-486 decodes what it is given) and with large seeds
    split by the words read ahead (max_seeds)."""
    from test_hashing import restate_candidates, restate_kwargs

    sp = small_problem
    out = str(tmp_path / "hashknobs")
    p = _params(files, out, hashing=True, **opts)
    api.FastSMC(p).run()
    got = gzip.open(out + ".1.1.FastSMC.ibd.gz", "rt").read()
    individuals = list(range(32))
    cands = restate_candidates(sp["haps"].alleles, sp["gen"], individuals, **restate_kwargs(opts))
    assert len(cands) >= 20
    if not opts.get("haploid", True):
        assert any(a == b for a, b, _, _ in cands)
    recs, pairs = _oracle_hashing_records(sp, sp["folded"], sp["model"], cands)
    ids = [f"1_{d + 1}" for d in individuals]
    want = O.format_ibd_text(recs, pairs, ids, ids, 1, sp["haps"].bp, sp["gen"])
    assert want.count("\n") > 10
    assert got == want


def test_binary_output_round_trip(files, small_problem, tmp_path):
    """BIN_OUT writes the .bibd.gz layout of HMM.cpp:383-401 / 1146-1176; read it back with BinaryDataReader and
    compare with the text run (ibd_score is fp32 in the binary file, double in the text file)."""
    out_b = str(tmp_path / "bin")
    out_t = str(tmp_path / "txt")
    api.FastSMC(_params(files, out_b, BIN_OUT=True)).run()
    api.FastSMC(_params(files, out_t)).run()
    text = gzip.open(out_t + ".1.1.FastSMC.ibd.gz", "rt").read().splitlines()
    rd = api.BinaryDataReader(out_b + ".1.1.FastSMC.bibd.gz")
    n = 0
    while rd.moreLinesInFile():
        line = rd.getNextLine()
        t = text[n].split("\t")
        b = line.toString().split("\t")
        assert b[:10] == t[:10] and b[11:] == t[11:]
        assert float(b[10]) == pytest.approx(float(t[10]), rel=1e-6)
        n += 1
    assert n == len(text) and n > 100


@pytest.mark.parametrize("K", [100, 300, 470, 530, 1040])
def test_fastsmc_run_wide_model_matches_oracle_text(small_problem, tmp_path, K):
    """The same end-to-end run with a 100-state model: routed to the 112-state member of the kernel family (12 ghost
    states) through the ordinary host path -- files, Data, HMM, FastSMC.run(), text output.  With 300 states: the
    wave-group kernel at four waves of 80 states; with 470: eight waves of 64; with 530: eight waves of 80 without landing zones;
    with 1040: beyond every register-resident kernel, decoded by the any-K kernel (fsmc_kernels_any.h)."""
    sp = dict(small_problem)
    sp["tables"] = synth.make_model_tables(K)
    root = str(tmp_path / "syn100")
    synth.write_haps_files(root, sp["haps"])
    used = np.unique(np.concatenate([[0.0], O.step_rows(sp["tables"].keys, sp["gen"])[1][1:]]))
    t = copy.copy(sp["tables"])
    sel = np.nonzero(np.isin(t.keys, used.astype(np.float32)))[0]
    t.keys, t.D, t.B, t.U, t.RR = t.keys[sel], t.D[sel], t.B[sel], t.U[sel], t.RR[sel]
    synth.write_decoding_quantities(root + ".decodingQuantities.gz", t)
    out = str(tmp_path / "res100")
    api.FastSMC(_params(root, out)).run()
    got = gzip.open(out + ".1.1.FastSMC.ibd.gz", "rt").read()
    want, n_pairs = _oracle_text(sp, list(range(32)), 1, 1)
    assert n_pairs == 2 * 32 * 32 - 32 and want.count("\n") > 50
    assert got == want
