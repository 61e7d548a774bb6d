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


def test_emission_preparation_ignores_other_threads_calling_rand(small_problem):
    """The reference seeds the PROCESS's rand() in Data's constructor and draws the emission preparation's seeds from it in
    HMM's (Data.cpp:62-70, 144-160).  Another thread that calls rand() in between shifts that sequence -- the HIP runtime
    does while it initialises, on the drivers' helper thread.  A Data object carries its own glibc-compatible generator:
    with a thread hammering libc's rand() the prepared model is still the oracle's (which calls the real generator)."""
    import ctypes
    import threading

    libc = ctypes.CDLL(None)
    stop = threading.Event()

    def hammer():
        while not stop.is_set():
            for _ in range(1000):
                libc.rand()

    t = threading.Thread(target=hammer)
    t.start()
    try:
        sp = small_problem
        for _ in range(3):
            data = api.Data.from_arrays(sp["haps"].alleles, sp["haps"].bp, sp["haps"].cm, True, True)
            hmm = api.HMM(data, api.decoding_quantities_from_tables(sp["tables"]), _params())
            _assert_same_model(hmm.preparedModel(), sp["model"])
    finally:
        stop.set()
        t.join()


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


def _assert_same_sequence_rows(pm_host: dict, pm_oracle):
    assert pm_host["sequence"] and pm_oracle.sequence
    for rows in ("gap_row_f", "site_row_f", "gap_row_b", "site_row_b"):
        for name in ("D", "B", "U", "RR"):
            got = pm_host[name][pm_host[rows][1:]]
            want = getattr(pm_oracle, name)[getattr(pm_oracle, rows)[1:]]
            np.testing.assert_array_equal(got, want, err_msg=f"{name}[{rows}]")
    np.testing.assert_array_equal(pm_host["hom"][1:], pm_oracle.hom[1:])


@pytest.mark.parametrize("fold", [True, False])
def test_prepared_model_sequence_mode(seq_problem, fold):
    """decodingSequence: CSFS / folded CSFS / classic emissions (HMM.cpp:183-252), per-site rows of the two
    transition steps and the homozygous emission between sites (HMM.cpp:752-770, 905-925)."""
    sp = seq_problem
    data = api.Data.from_arrays(sp["haps"].alleles, sp["haps"].bp, sp["haps"].cm, fold, True)
    np.testing.assert_array_equal(np.array(data.recRateAtMarker, np.float32),
                                  O.rec_rate_at_marker(sp["gen"], sp["haps"].bp))
    dq = api.decoding_quantities_from_tables(sp["tables"])
    p = _params(decodingSequence=True, decodingModeString="sequence", foldData=fold)
    hmm = api.HMM(data, dq, p)
    if fold:
        want = sp["model"]
    else:
        derived = sp["haps"].alleles.sum(axis=0).astype(np.int32)
        want = O.prepare_model(sp["tables"], sp["gen"], sp["haps"].bp, derived, 64, time=50, fold=False,
                               decoding_sequence=True)
    got = hmm.preparedModel()
    _assert_same_model(got, want)
    _assert_same_sequence_rows(got, want)


def test_sequence_mode_files_round_trip(seq_problem, tmp_path):
    """HomozygousEmissions / CSFS / FoldedCSFS / ClassicEmission sections through the file parser
    (DecodingQuantities.cpp:283-284, 337-343), ASMC-mode readers (plink map, Data.cpp:162-210)."""
    sp = seq_problem
    root = str(tmp_path / "seq")
    synth.write_haps_files(root, sp["haps"], fastsmc_map=False)
    synth.write_decoding_quantities(root + ".decodingQuantities.gz", sp["tables"])
    p = api.DecodingParams(root, root + ".decodingQuantities.gz", decodingModeString="sequence")
    p.useKnownSeed = True
    assert p.decodingSequence and p.foldData
    dq = api.DecodingQuantities(root + ".decodingQuantities.gz")
    hom = dq.homozygousEmissionMap
    assert len(hom) == sp["tables"].homozygous_keys.size
    np.testing.assert_array_equal(np.array(hom[1000], np.float32),
                                  sp["tables"].homozygous[list(sp["tables"].homozygous_keys).index(1000)])
    data = api.Data(p)
    hmm = api.HMM(data, p)
    got = hmm.preparedModel()
    # the ASMC-mode map reader computes positions and rates in fp32 (Data.cpp:186-195)
    gen = np.array(data.geneticPositions, np.float32)
    phys = np.array(data.physicalPositions, np.int64)
    rate = np.zeros(gen.size, np.float32)
    rate[1:] = (gen[1:] - gen[:-1]) / (phys[1:] - phys[:-1]).astype(np.float32)
    np.testing.assert_array_equal(np.array(data.recRateAtMarker, np.float32), rate)
    _, derived, _ = synth.fold_and_pack(sp["haps"].alleles)
    want = O.prepare_model(sp["tables"], gen, phys, derived, 64, time=p.time, decoding_sequence=True, rec_rate=rate,
                           no_conditional_age_estimates=p.noConditionalAgeEstimates)
    _assert_same_model(got, want)
    _assert_same_sequence_rows(got, want)


def test_expected_coal_times_file_is_read_like_the_reference(tmp_path):
    """HMM.cpp:43-61, 1736-1748: the second column of the intervals file replaces the decoding quantities' expected
    times in ASMC mode; a line without three fields is an error (the reference exits with this message); FastSMC mode
    never reads the file.  No GPU involved: the engine opens at the first decode."""
    import numpy as np
    import pytest
    from fastsmc_amd import api, synth

    tables = synth.make_model_tables(12)
    haps = synth.make_haps(64, 60, seed=3)
    data = api.Data.from_arrays(haps.alleles, haps.bp, haps.cm, True, True)
    dq = api.decoding_quantities_from_tables(tables)
    K = len(tables.expected_times)
    good = str(tmp_path / "good.intervalsInfo")
    custom = np.linspace(5.0, 5000.0, K).astype(np.float32)
    with open(good, "w") as f:
        for k in range(K):
            f.write(f"{k}\t{float(custom[k])!r}\t{k + 1}\n")
    p = api.DecodingParams()
    p.FastSMC = False
    p.expectedCoalTimesFile = good
    np.testing.assert_array_equal(np.array(api.HMM(data, dq, p).getExpectedCoalTimes(), np.float32), custom)
    p.expectedCoalTimesFile = str(tmp_path / "missing")   # not a regular file: the decoding quantities' times
    np.testing.assert_array_equal(np.array(api.HMM(data, dq, p).getExpectedCoalTimes(), np.float32),
                                  np.asarray(tables.expected_times, np.float32))
    bad = str(tmp_path / "bad.intervalsInfo")
    open(bad, "w").write("0\t1.5\t2\n3 4\n")
    p.expectedCoalTimesFile = bad
    with pytest.raises(RuntimeError, match="intervalStart"):
        api.HMM(data, dq, p)
    short = str(tmp_path / "short.intervalsInfo")
    open(short, "w").write("0\t1.5\t2\n")
    p.expectedCoalTimesFile = short
    with pytest.raises(RuntimeError, match="intervals"):
        api.HMM(data, dq, p)
    p.FastSMC = True
    p.expectedCoalTimesFile = bad
    api.HMM(data, dq, p)  # FastSMC mode does not look at it


def test_expected_coal_times_from_the_reference_intervals_file():
    """The reference's own input of `expectedCoalTimesFile` -- FILES/DECODING_QUANTITIES/30-100-2000.intervalsInfo, 69
    lines "intervalStart <tab> expectedCoalescentTime <tab> intervalEnd", committed as tests/golden/30-100-2000.intervalsInfo
    -- read through the product's reader (HMM.cpp:43-61): the 69 values are the second column, each parsed as the
    reference's StringUtils::stof does (stold, then one rounding to float: StringUtils.cpp:36-39), they lie inside their
    intervals, and an ASMC-mode HMM of a 69-state model hands them out as its expected coalescence times."""
    import numpy as np
    from fastsmc_amd import api, synth

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "30-100-2000.intervalsInfo")
    rows = [ln.split() for ln in open(path)]
    assert len(rows) == 69 and all(len(r) == 3 for r in rows)
    want = np.array([np.float32(np.longdouble(r[1])) for r in rows], np.float32)
    starts = np.array([float(r[0]) for r in rows])
    ends = np.array([float(r[2]) for r in rows])  # (the last interval ends at "Infinity")
    assert starts[0] == 0.0 and np.array_equal(starts[1:], ends[:-1]) and np.isinf(ends[-1])
    assert np.all(want > starts) and np.all(want[:-1] < ends[:-1])
    tables = synth.make_model_tables(69)
    haps = synth.make_haps(64, 60, seed=3)
    data = api.Data.from_arrays(haps.alleles, haps.bp, haps.cm, True, True)
    dq = api.decoding_quantities_from_tables(tables)
    p = api.DecodingParams()
    p.FastSMC = False
    p.expectedCoalTimesFile = path
    got = np.array(api.HMM(data, dq, p).getExpectedCoalTimes(), np.float32)
    np.testing.assert_array_equal(got, want)
    assert not np.array_equal(got, np.asarray(tables.expected_times, np.float32))  # (the file's, not the model's)


def test_bench_unpacks_the_folded_alleles_it_gives_the_cpu_baseline():
    """bench.py's cpu_baseline decodes the first pairs of the GPU's own work list on the host: the folded alleles it
    hands the oracle come out of the packed matrix the GPU decodes from (`folded_rows_from_bits`) -- they must be the
    folded alleles of the cohort, whatever the number of sites modulo 64."""
    import numpy as np
    import bench
    from fastsmc_amd import synth

    for n_sites in (64, 100, 129):
        pm, bits, haps, _ = bench.build_problem(64, n_sites, 12, seed=5)
        _, _, flipped = synth.fold_and_pack(haps.alleles)
        folded = np.where(flipped[None, :], 1 - haps.alleles, haps.alleles).astype(np.uint8)
        rows = np.array([0, 3, 17, 63])
        np.testing.assert_array_equal(bench.folded_rows_from_bits(bits, rows, n_sites), folded[rows])


def test_ibd_text_of_many_records_is_the_text_of_few(small_problem, tmp_path):
    """A flush of many records is formatted by several threads and compressed into gzip members of their own
    (host/hmm.cpp: formatIbdRecords, putIbdText); the text a reader sees must be what the small-flush path -- one stream,
    gzwrite -- writes for the same records.  No device: writeIbdRecordArrays only formats and writes."""
    import gzip

    sp = small_problem
    data = api.Data.from_arrays(sp["haps"].alleles, sp["haps"].bp, sp["haps"].cm, True, True)
    hmm = api.HMM(data, api.decoding_quantities_from_tables(sp["tables"]), _params())
    rng = np.random.default_rng(5)
    n, S, H = 30000, sp["haps"].alleles.shape[1], sp["haps"].alleles.shape[0]
    start = rng.integers(0, S - 1, n)
    cols = dict(hap_a=rng.integers(0, H, n).astype(np.uint32), hap_b=rng.integers(0, H, n).astype(np.uint32),
                start=start.astype(np.int32), end=np.minimum(start + rng.integers(0, 200, n), S - 1).astype(np.int32),
                prob=rng.random(n).astype(np.float32) * 50, post_mean=rng.random(n).astype(np.float32) * 3000,
                map=rng.integers(0, 69, n).astype(np.float32))
    big = str(tmp_path / "big.ibd.gz")
    hmm.writeIbdRecordArrays(big, **cols)
    text = gzip.open(big, "rt").read()
    assert text.count("\n") == n
    few = []
    for lo in range(0, n, 3000):  # (slices of 300 records: below the threshold of either mechanism)
        name = str(tmp_path / "few.ibd.gz")
        hmm.writeIbdRecordArrays(name, **{k: v[lo:lo + 300] for k, v in cols.items()})
        few.append((lo, gzip.open(name, "rt").read()))
    lines = text.splitlines(keepends=True)
    for lo, t in few:
        assert "".join(lines[lo:lo + 300]) == t
    # every reader of gzip files sees one stream: zlib's own too
    import subprocess
    assert subprocess.run(["zcat", big], capture_output=True, check=True).stdout.decode() == text
