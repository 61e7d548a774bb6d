"""The identification step on the GPU (fsmc_identify; scope row f1 -- FastSMC.cpp:118-235 with HASHING/SeedHash.hpp,
ExtendHash.hpp, Match.hpp): every pair of the job is a lane's state machine over its word equalities.

Checkers, all integer-exact (the candidate list must be EQUAL, order included):
  * tests/test_hashing.py::restate_candidates -- the reference's formulation (a seed table per word, a map of open
    matches, flushes), written independently in Python;
  * fastsmc_amd.api.hashingCandidates -- the host C++ restatement of the same (the product path no longer calls it).
Cases: the options the reference exposes (min_m, gap, skip, min_maf, folding off), job windows (jobs = 4, 9: per job
equal to the restatement, over all jobs a partition of the single-job list), ragged shapes through the raw C ABI
(haplotypes and words that do not fill a tile, fewer than two haplotypes, no complete word), low-complexity words,
the overflow protocol, and a cohort big enough to fill the machine."""
import numpy as np
import pytest

from fastsmc_amd import api, capi, synth
from oracle import oracle as O
from test_hashing import OTHER_KNOBS, _params, restate_candidates, restate_kwargs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hash_files(tmp_path_factory):
    haps = synth.make_haps(80, 1500, seed=11, cm_per_mb=25.0, switch_per_cm=0.5, noise=1e-3)
    root = str(tmp_path_factory.mktemp("hash") / "syn")
    synth.write_haps_files(root, haps)
    return root, haps


@pytest.mark.parametrize("opts", [dict(), dict(min_m=0.5, gap=0), dict(min_m=2.0, gap=3), dict(skip=0.9),
                                  dict(min_maf=0.05, min_m=0.8), dict(foldData=False), dict(skip=0.3, gap=2, min_m=0.3)])
def test_device_candidates_equal_the_restatement_and_the_host(hash_files, opts):
    root, haps = hash_files
    p = _params(root, **opts)
    data = api.Data(p)
    got = [tuple(c) for c in api.hashingCandidatesDevice(data, p)]
    kw = {k: v for k, v in opts.items() if k != "foldData"}
    want = restate_candidates(haps.alleles, (haps.cm / 100.0).astype(np.float32), list(range(40)), **kw)
    assert got == want
    assert got == [tuple(c) for c in api.hashingCandidates(data, p)]
    if not opts.get("skip"):
        assert len(want) > 20


@pytest.mark.parametrize("opts", OTHER_KNOBS)
def test_word_size_individual_pairs_and_split_seeds_on_the_device(hash_files, opts):
    """hashingWordSize != 64, haploid = false (matches keyed by individual pairs) and max_seeds != 0 (large seeds split
    by the words read ahead: fsmc_identify_seeds.hip + the general match kernel) -- equal to both restatements."""
    root, haps = hash_files
    p = _params(root, **opts)
    data = api.Data(p)
    got = [tuple(c) for c in api.hashingCandidatesDevice(data, p)]
    want = restate_candidates(haps.alleles, (haps.cm / 100.0).astype(np.float32), list(range(40)),
                              **restate_kwargs(opts))
    assert got == want
    assert got == [tuple(c) for c in api.hashingCandidates(data, p)]
    assert len(want) > 10


@pytest.mark.parametrize("jobs", [4])
@pytest.mark.parametrize("opts", [dict(haploid=False), dict(max_seeds=2, constReadAhead=4),
                                  dict(max_seeds=2, haploid=False, hashingWordSize=32)])
def test_other_knobs_with_job_windows_on_the_device(hash_files, jobs, opts):
    root, haps = hash_files
    gen = (haps.cm / 100.0).astype(np.float32)
    total = 0
    for j in range(1, jobs + 1):
        p = _params(root, min_m=0.5, jobs=jobs, jobInd=j, **opts)
        data = api.Data(p)
        individuals = O.job_individuals(40, jobs, j)
        got = [tuple(c) for c in api.hashingCandidatesDevice(data, p)]
        assert got == restate_candidates(haps.alleles, gen, individuals, jobs=jobs, job_ind=j, min_m=0.5,
                                         **restate_kwargs(opts))
        total += len(got)
    assert total > 20


@pytest.mark.parametrize("jobs", [4, 9])
def test_job_windows_on_the_device(hash_files, jobs):
    root, haps = hash_files
    gen = (haps.cm / 100.0).astype(np.float32)
    p1 = _params(root, min_m=0.8)
    whole = {tuple(c) for c in api.hashingCandidatesDevice(api.Data(p1), p1)}
    seen = []
    for j in range(1, jobs + 1):
        p = _params(root, min_m=0.8, jobs=jobs, jobInd=j)
        data = api.Data(p)
        individuals = O.job_individuals(40, jobs, j)
        got = [tuple(c) for c in api.hashingCandidatesDevice(data, p)]
        assert got == restate_candidates(haps.alleles, gen, individuals, jobs=jobs, job_ind=j, min_m=0.8)
        rows = [2 * d + h for d in individuals for h in (0, 1)]
        seen += [tuple(sorted((rows[a], rows[b]))) + (f, t) for a, b, f, t in got]
    assert len(seen) == len(set(seen))
    assert set(seen) == whole and len(whole) > 20


def _pack_words(alleles, word_size=64):
    n, S = alleles.shape
    W = S // word_size
    bits = alleles[:, :W * word_size].reshape(n, W, word_size).astype(np.uint64)
    return (bits << np.arange(word_size, dtype=np.uint64)[None, None, :]).sum(axis=2, dtype=np.uint64)


def _raw(ctx, alleles, gen, **kw):
    words = _pack_words(alleles, kw.get("word_size", 64))
    ids = np.arange(alleles.shape[0], dtype=np.uint32)
    rec = ctx.identify(words, ids, gen, **kw)
    return [(int(r["hap_a"]), int(r["hap_b"]), int(r["from"]), int(r["to"])) for r in rec], rec


@pytest.mark.parametrize("n_hap,S", [(2, 64), (6, 200), (34, 64 * 33 + 5), (70, 64 * 40), (130, 64 * 7 + 63)])
def test_ragged_shapes_through_the_c_abi(n_hap, S):
    haps = synth.make_haps(n_hap, S, seed=n_hap + S, cm_per_mb=40.0, switch_per_cm=0.3, noise=2e-3, n_founders=5)
    gen = (haps.cm / 100.0).astype(np.float32)
    ctx = capi.Context(0)
    for kw in (dict(min_m=0.0, gap=0), dict(min_m=0.2, gap=1), dict(min_m=0.0, gap=2, skip=0.5)):
        got, rec = _raw(ctx, haps.alleles, gen, **kw)
        want = restate_candidates(haps.alleles, gen, list(range(n_hap // 2)), **kw)
        assert got == want, kw
        # the order is the documented one: by flush word, then by pair key
        key = rec["flush_word"].astype(np.int64) * (n_hap * n_hap) + rec["hap_a"].astype(np.int64) * n_hap + rec["hap_b"]
        assert np.all(np.diff(key) > 0)
    ctx.close()


@pytest.mark.parametrize("n_hap,S", [(2, 64), (6, 200), (34, 64 * 33 + 5), (70, 64 * 40), (130, 64 * 7 + 63)])
def test_ragged_shapes_with_the_other_knobs(n_hap, S):
    """The general match kernel and the seed splitting on shapes that do not fill tiles or chunks; read-ahead windows
    that reach past the last word; seeds of every size (max_seeds = 1 splits every seed that is not a singleton)."""
    haps = synth.make_haps(n_hap, S, seed=n_hap + S, cm_per_mb=40.0, switch_per_cm=0.3, noise=2e-3, n_founders=5)
    gen = (haps.cm / 100.0).astype(np.float32)
    ctx = capi.Context(0)
    for kw in (dict(min_m=0.0, gap=0, haploid=False), dict(min_m=0.1, gap=1, word_size=20),
               dict(min_m=0.0, gap=1, max_seeds=1, read_ahead=2, word_size=16),
               dict(min_m=0.0, gap=2, max_seeds=2, read_ahead=32, word_size=8, skip=0.02),
               dict(min_m=0.05, gap=0, max_seeds=3, haploid=False, word_size=32, read_ahead=10)):
        got, rec = _raw(ctx, haps.alleles, gen, **kw)
        want = restate_candidates(haps.alleles, gen, list(range(n_hap // 2)), **kw)
        assert got == want, kw
        key = rec["flush_word"].astype(np.int64) * (n_hap * n_hap) + rec["hap_a"].astype(np.int64) * n_hap + rec["hap_b"]
        assert np.all(np.diff(key) > 0)
    ctx.close()


def test_the_other_knobs_are_checked():
    ctx = capi.Context(0)
    gen = np.linspace(0, 0.01, 400).astype(np.float32)
    w = np.zeros((4, 4), np.uint64)
    ids = np.arange(4, dtype=np.uint32)
    for bad in (dict(word_size=0), dict(word_size=65), dict(max_seeds=2, read_ahead=0), dict(max_seeds=2, read_ahead=33)):
        with pytest.raises(capi.FsmcError):
            ctx.identify(w, ids, gen, **bad)
    with pytest.raises(capi.FsmcError):   # individuals are rows 2k, 2k+1
        ctx.identify(np.zeros((5, 4), np.uint64), np.arange(5, dtype=np.uint32), gen, haploid=False)
    assert ctx.identify(w, ids, gen, max_seeds=-1, min_m=0.0).size == 6   # (never splits: compared as unsigned long)
    ctx.close()


def test_degenerate_inputs():
    ctx = capi.Context(0)
    gen = np.linspace(0, 0.01, 200).astype(np.float32)
    one = np.zeros((1, 3), np.uint64)
    assert ctx.identify(one, np.zeros(1, np.uint32), gen).size == 0            # fewer than two haplotypes
    assert ctx.identify(np.zeros((4, 0), np.uint64), np.arange(4, dtype=np.uint32), gen).size == 0  # no complete word
    with pytest.raises(capi.FsmcError):                                          # more words than sites
        ctx.identify(np.zeros((4, 4), np.uint64), np.arange(4, dtype=np.uint32), gen)
    # identical haplotypes: one candidate per pair, the whole range, reported at the end
    same = np.full((5, 3), 0x0123456789ABCDEF, np.uint64)
    rec = ctx.identify(same, np.arange(5, dtype=np.uint32), gen, min_m=0.0)
    assert rec.size == 10 and set(rec["flush_word"]) == {3} and set(rec["from"]) == {0} and set(rec["to"]) == {191}
    # ... and none of them when every word is low-complexity (one distinct value / 5 haplotypes = 0.2 <= skip)
    assert ctx.identify(same, np.arange(5, dtype=np.uint32), gen, min_m=0.0, skip=0.2).size == 0
    ctx.close()


def test_low_complexity_words_carry_open_matches():
    """Word 1 has two distinct values among six haplotypes (2/6 <= skip = 0.4): it is not compared, and the match a
    pair opened on word 0 is carried over it (ExtendHash.hpp:100-104) -- also for a pair that DIFFERS on word 1."""
    rng = np.random.default_rng(3)
    alleles = rng.integers(0, 2, size=(6, 64 * 4), dtype=np.uint8)
    alleles[1, :64] = alleles[0, :64]          # pair (0, 1) matches on word 0
    alleles[:, 64:128] = 0
    alleles[0, 64] = 1                         # word 1: two distinct values, (0, 1) differ
    alleles[1, 128:192] = alleles[0, 128:192]  # and match again on word 2
    gen = (np.arange(alleles.shape[1]) * 1e-4).astype(np.float32)
    ctx = capi.Context(0)
    got, _ = _raw(ctx, alleles, gen, min_m=0.0, gap=0, skip=0.4)
    ctx.close()
    assert got == restate_candidates(alleles, gen, [0, 1, 2], min_m=0.0, gap=0, skip=0.4)
    assert (0, 1, 0, 191) in got


def test_overflow_protocol_and_a_machine_filling_cohort():
    """2048 haplotypes x 128 words: 2.1 M pairs on 2080 tiles.  The first call's buffer is too small on purpose
    (capi.Context.identify starts at 4 * n_haps): FSMC_EOVERFLOW reports the count and the second call is complete.
    Checked against the host restatement (C++), which the small cases above tie to the Python one."""
    haps = synth.make_haps_blocked(2048, 64 * 128, seed=5, cm_per_mb=30.0, switch_per_cm=0.05, noise=5e-4, n_founders=40)
    gen = (haps.cm / 100.0).astype(np.float32)
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        root = d + "/big"
        synth.write_haps_files(root, haps)
        p = _params(root, min_m=1.0)
        data = api.Data(p)
        want = [tuple(c) for c in api.hashingCandidates(data, p)]
    ctx = capi.Context(0)
    got, rec = _raw(ctx, haps.alleles, gen, min_m=1.0)
    ms = ctx.last_kernel_ms()
    ctx.close()
    assert len(want) > 4 * 2048  # the first buffer was too small
    assert got == want
    assert ms > 0


def test_split_seeds_and_individual_pairs_on_a_machine_filling_cohort():
    """The same cohort shape with large seeds (40 founders: seeds of ~50 haplotypes, max_seeds = 20 splits them, some
    several times) and with matches keyed by individual pairs, against the host restatement (C++)."""
    haps = synth.make_haps_blocked(1024, 32 * 160, seed=9, cm_per_mb=30.0, switch_per_cm=0.05, noise=5e-4, n_founders=40)
    gen = (haps.cm / 100.0).astype(np.float32)
    import tempfile
    ctx = capi.Context(0)
    with tempfile.TemporaryDirectory() as d:
        root = d + "/big"
        synth.write_haps_files(root, haps)
        plain = None
        for opts in (dict(), dict(max_seeds=20, constReadAhead=8), dict(haploid=False, max_seeds=20)):
            p = _params(root, min_m=1.0, hashingWordSize=32, **opts)
            data = api.Data(p)
            want = [tuple(c) for c in api.hashingCandidates(data, p)]
            kw = dict(word_size=32, haploid=opts.get("haploid", True), max_seeds=opts.get("max_seeds", 0),
                      read_ahead=opts.get("constReadAhead", 10))
            got, _ = _raw(ctx, haps.alleles, gen, min_m=1.0, **kw)
            assert len(want) > 1000
            assert got == want, opts
            if not opts:
                plain = want
            elif opts.get("haploid", True):
                assert want != plain   # the splitting changed the list
    ctx.close()
