"""Row a10's file outputs (HMM::writePerPairOutput, HMM.cpp:1360-1458): in ASMC mode `setWritePerPairPosteriorMean` /
`setWritePerPairMap` (HMM.hpp:287, 293) make decodeAll write <out>.perPairPosteriorMeans.gz / .perPairMAP.gz -- one row
per decoded pair, streamed batch by batch through Eigen's `format` (HMM.cpp:1412-1420, HMM.hpp:154) -- and
`DecodingParams.expectedCoalTimesFile` replaces the decoding quantities' expected times in the posterior means
(HMM.cpp:43-61, 1736-1748).  Against the oracle's per-pair consumer formatted the same way."""
import copy
import gzip

import numpy as np
import pytest

from fastsmc_amd import api, synth
from oracle import oracle as O
from test_gpu_modes import _write_files

pytestmark = pytest.mark.gpu


def _oracle_model(sp, data, p):
    gen = np.array(data.geneticPositions, np.float32)
    _, derived, _ = synth.fold_and_pack(sp["haps"].alleles)
    return O.prepare_model(sp["tables"], gen, sp["haps"].bp, derived, 64, time=p.time,
                           no_conditional_age_estimates=False)


def _batches(sp, pm, pairs, batch):
    folded = sp["folded"]
    for b0 in range(0, len(pairs), batch):
        chunk = pairs[b0:b0 + batch]
        ob = np.stack([folded[x] ^ folded[y] for x, y in chunk])
        hb = np.stack([folded[x] & folded[y] for x, y in chunk])
        post, _ = O.decode_batch(pm, ob, hb, 0, pm.S)
        yield chunk, post


@pytest.mark.parametrize("batch", [64, 40])
def test_decode_all_writes_the_per_pair_files(small_problem, tmp_path, batch):
    sp = small_problem
    root = str(tmp_path / f"pp{batch}")
    _write_files(sp, root)
    p = api.DecodingParams(root, root + ".decodingQuantities.gz")
    p.useKnownSeed = True
    p.batchSize = batch
    p.outFileRoot = root
    p.jobs, p.jobInd = 4, 2
    data = api.Data(p)
    hmm = api.HMM(data, p)
    hmm.setWritePerPairPosteriorMean(True)
    hmm.setWritePerPairMap(True)
    hmm.decodeAll(p.jobs, p.jobInd)
    got_mean = gzip.open(root + ".perPairPosteriorMeans.gz", "rt").read()
    got_map = gzip.open(root + ".perPairMAP.gz", "rt").read()
    pm = _oracle_model(sp, data, p)
    pairs = O.enumerate_all_pairs(32, 4, 2)
    assert len(pairs) % batch != 0 and len(pairs) > 2 * batch  # a ragged last batch, several batches
    want_mean, want_map = "", ""
    for chunk, post in _batches(sp, pm, pairs, batch):
        wmean, wmap, _ = O.per_pair_output(pm, post, len(chunk))
        want_mean += O.eigen_format_rows(wmean)  # (the reference streams matrix after matrix: no newline in between)
        want_map += O.eigen_format_rows(wmap)
    assert got_mean == want_mean
    assert got_map == want_map
    # the values themselves, whatever the text: parsed back batch by batch (nothing separates a batch's last value from
    # the next batch's first), the means are the oracle's to the 6 printed digits
    def parse(text, digits):
        vals, at = [], 0
        for chunk, post in _batches(sp, pm, pairs, batch):
            wmean = O.per_pair_output(pm, post, len(chunk))[0]
            n = sum(len("%.*g" % (digits, float(x))) for x in wmean.reshape(-1)) + wmean.size - 1  # values + separators
            vals.append(np.array(text[at:at + n].split(), np.float64))
            at += n
        assert at == len(text)
        return np.concatenate(vals)

    all_mean = np.concatenate([O.per_pair_output(pm, post, len(chunk))[0].reshape(-1) for chunk, post in _batches(sp, pm, pairs, batch)])
    np.testing.assert_allclose(parse(got_mean, 6), all_mean, rtol=5e-6)
    # Eigen 5 / master print max_digits10 = 9 digits where Eigen 3.4 prints 6: the environment switch writes that file,
    # whose values round-trip to the float32 means bit for bit
    import os
    os.environ["FSMC_EIGEN_FULL_PRECISION_DIGITS"] = "9"
    try:
        hmm.decodeAll(p.jobs, p.jobInd)
    finally:
        del os.environ["FSMC_EIGEN_FULL_PRECISION_DIGITS"]
    nine = gzip.open(root + ".perPairPosteriorMeans.gz", "rt").read()
    assert nine != got_mean
    np.testing.assert_array_equal(parse(nine, 9).astype(np.float32), all_mean.astype(np.float32))
    # one row per pair, minus the joins at the batch boundaries
    n_batches = (len(pairs) + batch - 1) // batch
    assert got_map.count("\n") == len(pairs) - n_batches
    # a second decode re-opens (truncates) the files (HMM::resetDecoding, HMM.cpp:259-271)
    hmm.decodeAll(p.jobs, p.jobInd)
    assert gzip.open(root + ".perPairMAP.gz", "rt").read() == want_map
    # without the switches nothing is written
    hmm.setWritePerPairPosteriorMean(False)
    hmm.setWritePerPairMap(False)
    import os

    os.remove(root + ".perPairMAP.gz")
    hmm.decodeAll(p.jobs, p.jobInd)
    assert not os.path.exists(root + ".perPairMAP.gz")


def test_expected_coal_times_file_changes_the_means(small_problem, tmp_path):
    sp = small_problem
    root = str(tmp_path / "ect")
    _write_files(sp, root)
    t = sp["tables"]
    K = len(t.expected_times)
    custom = (np.asarray(t.expected_times, np.float64) * 1.75 + 3.0).astype(np.float32)
    intervals = str(tmp_path / "custom.intervalsInfo")
    with open(intervals, "w") as f:
        for k in range(K):
            f.write(f"{float(t.discretization[k])!r}\t{float(custom[k])!r}\t{float(t.discretization[k + 1])!r}\n")

    def run(ect_file):
        p = api.DecodingParams(root, root + ".decodingQuantities.gz", root, 1, 1, "array", False, True, False, False,
                               0.0, False, True, False, ect_file, False, True)
        p.doPerPairMAP = True
        p.useKnownSeed = True
        asmc = api.ASMC(p)
        a, b = [1, 2, 3, 10], [2, 3, 4, 11]
        asmc.decodePairs(a, b, False, False, True, True)
        return p, asmc.get_copy_of_results(), a, b

    p0, base, a, b = run("")
    p1, res, _, _ = run(intervals)
    assert p1.doPerPairPosteriorMean  # (DecodingParams.cpp:491-493)
    data = api.Data(p1)
    hmm = api.HMM(data, p1)
    np.testing.assert_array_equal(np.array(hmm.getExpectedCoalTimes(), np.float32), custom)
    pm = copy.copy(_oracle_model(sp, data, p1))
    pm.exp_times = custom
    folded = sp["folded"]
    ob = np.stack([folded[x] ^ folded[y] for x, y in zip(a, b)])
    hb = np.stack([folded[x] & folded[y] for x, y in zip(a, b)])
    post, _ = O.decode_batch(pm, ob, hb, 0, pm.S)
    wmean, wmap, _ = O.per_pair_output(pm, post, len(a))
    np.testing.assert_array_equal(res.per_pair_posterior_means, wmean)
    np.testing.assert_array_equal(res.per_pair_MAPs, wmap)
    assert not np.array_equal(res.per_pair_posterior_means, base.per_pair_posterior_means)
    np.testing.assert_array_equal(res.per_pair_MAPs, base.per_pair_MAPs)  # the MAP state does not use the times
