"""The product's host side (C++: Data, DecodingQuantities, HMM constructor) against the oracle's restatement
of the same preparation steps (HMM.cpp:65-127, 159-256; Data.cpp) -- bit-identical tables, no GPU needed."""
import os

import numpy as np
import pytest

from fastsmc_amd import api, synth
from oracle import oracle as O


def _params(**kw):
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
    for k, v in kw.items():
        setattr(p, k, v)
    return p


FIELDS = ("pi", "col_ratios", "exp_times", "D", "B", "U", "RR", "e1", "e0m1", "e2m0")


def _assert_same_model(pm_host: dict, pm_oracle):
    assert pm_host["K"] == pm_oracle.K and pm_host["S"] == pm_oracle.S
    assert pm_host["state_threshold"] == pm_oracle.state_threshold
    assert pm_host["age_threshold"] == pm_oracle.age_threshold
    assert np.float32(pm_host["probability_threshold"]) == pm_oracle.probability_threshold
    # row numbering may differ (first-use order vs sorted); compare the rows each site step selects
    for name in ("D", "B", "U", "RR"):
        got = pm_host[name][pm_host["step_row"][1:]]
        want = getattr(pm_oracle, name)[pm_oracle.step_row[1:]]
        np.testing.assert_array_equal(got, want, err_msg=name)
    for name in ("pi", "col_ratios", "exp_times", "e1", "e0m1", "e2m0"):
        np.testing.assert_array_equal(pm_host[name], getattr(pm_oracle, name), err_msg=name)


def test_prepared_model_from_arrays_matches_oracle(small_problem):
    sp = small_problem
    data = api.Data.from_arrays(sp["haps"].alleles, sp["haps"].bp, sp["haps"].cm, True, True)
    np.testing.assert_array_equal(data.packed_bits(), sp["bits"])
    np.testing.assert_array_equal(np.array(data.geneticPositions, np.float32), sp["gen"])
    dq = api.decoding_quantities_from_tables(sp["tables"])
    hmm = api.HMM(data, dq, _params())
    _assert_same_model(hmm.preparedModel(), sp["model"])


@pytest.mark.parametrize("opts", [dict(noConditionalAgeEstimates=False, time=120),
                                  dict(skipCSFSdistance=float("inf")), dict(skipCSFSdistance=0.0005)])
def test_prepared_model_options(small_problem, opts):
    sp = small_problem
    data = api.Data.from_arrays(sp["haps"].alleles, sp["haps"].bp, sp["haps"].cm, True, True)
    dq = api.decoding_quantities_from_tables(sp["tables"])
    hmm = api.HMM(data, dq, _params(**opts))
    _, derived, _ = synth.fold_and_pack(sp["haps"].alleles)
    want = O.prepare_model(sp["tables"], sp["gen"], sp["haps"].bp, derived, 64, time=opts.get("time", 50),
                           no_conditional_age_estimates=opts.get("noConditionalAgeEstimates", True),
                           skip_csfs_distance=opts.get("skipCSFSdistance", 0.0))
    _assert_same_model(hmm.preparedModel(), want)


def test_file_readers_round_trip(small_problem, tmp_path):
    """.hap.gz/.samples/.map + .decodingQuantities.gz written by synth, read by the C++ host
    (FastSMC-mode readers, Data.cpp:98-141, 397-565; parser DecodingQuantities.cpp:60-345)."""
    sp = small_problem
    root = str(tmp_path / "syn")
    synth.write_haps_files(root, sp["haps"])
    used = np.unique(np.concatenate([[0.0], O.step_rows(sp["tables"].keys, sp["gen"])[1][1:]]))
    import copy
    t = copy.copy(sp["tables"])
    sel = np.nonzero(np.isin(t.keys, used.astype(np.float32)))[0]
    t.keys, t.D, t.B, t.U, t.RR = t.keys[sel], t.D[sel], t.B[sel], t.U[sel], t.RR[sel]
    synth.write_decoding_quantities(root + ".decodingQuantities.gz", t)
    p = api.DecodingParams(in_dir=root, decoding_quants=root + ".decodingQuantities.gz", out_dir=str(tmp_path / "out"),
                           FastSMC=True)
    p.hashing = False
    p.useKnownSeed = True
    data = api.Data(p)
    assert data.sites == sp["model"].S and data.sampleSize == 32 and data.chrNumber == 1
    np.testing.assert_array_equal(data.packed_bits(), sp["bits"])
    np.testing.assert_array_equal(np.array(data.physicalPositions), sp["haps"].bp)
    np.testing.assert_array_equal(np.array(data.geneticPositions, np.float32), sp["gen"])
    dq = api.DecodingQuantities(root + ".decodingQuantities.gz")
    assert dq.states == 69 and dq.CSFSSamples == sp["tables"].csfs_samples
    np.testing.assert_array_equal(np.array(dq.expectedTimes, np.float32), sp["tables"].expected_times)
    hmm = api.HMM(data, p)
    _assert_same_model(hmm.preparedModel(), sp["model"])


def test_decoding_quantities_validation(tmp_path):
    """test_decoding_quantities.cpp:24-44 of the reference: a file must exist and start with TransitionType."""
    with pytest.raises(RuntimeError, match="does not exist"):
        api.DecodingQuantities(str(tmp_path / "nope.gz"))
    bad = tmp_path / "bad.txt"
    bad.write_text("This is not a decoding quantities file\nsecond line\n")  # data/decoding_quantities_bad.txt
    with pytest.raises(RuntimeError, match="does not seem to contain the correct information"):
        api.DecodingQuantities(str(bad))


def test_params_defaults_match_reference():
    """DecodingParams.cpp:56-73 (FastSMC defaults) and test_decoding_params.cpp:22-62 (mode resolution)."""
    # NB: positional (str, str, str, True) binds the 17-argument ASMC overload first, as in the reference
    # (pybind.cpp:122-144) -- the FastSMC-defaults constructor is reached by keyword.
    p = api.DecodingParams(in_dir="in", decoding_quants="dq", out_dir="out", FastSMC=True)
    assert (p.batchSize, p.time, p.hashing, p.FastSMC, p.foldData) == (32, 50, True, True, True)
    assert p.noConditionalAgeEstimates and p.doPerPairPosteriorMean and p.doPerPairMAP and p.outputIbdSegmentLength
    assert abs(p.min_m - 1.5) < 1e-7 and p.skipCSFSdistance == 0.0
    q = api.DecodingParams("in", "dq")
    assert q.decodingModeOverall == api.DecodingModeOverall.array and q.decodingMode == api.DecodingMode.arrayFolded
    q = api.DecodingParams("in", "dq", decodingModeString="sequence")
    assert q.decodingModeOverall == api.DecodingModeOverall.sequence and q.decodingSequence
    q = api.DecodingParams("in", "dq", decodingModeString="array", useAncestral=True)
    assert q.decodingMode == api.DecodingMode.array and not q.foldData
    with pytest.raises(RuntimeError):
        api.DecodingParams("in", "dq", decodingModeString="banana")
