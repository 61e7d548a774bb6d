"""The identification pre-filter (scope row f1; reference FastSMC.cpp:118-235, HASHING/*.hpp): known answers of the
reference's own unit tests (TESTS/test_hashing.cpp) and the candidate stream against an independent restatement
(dictionary of seeds per word, dictionary of open matches) on synthetic haplotypes, with and without job windows."""
import copy
from collections import defaultdict

import numpy as np
import pytest

from fastsmc_amd import api, synth
from oracle import oracle as O


# ---------------------------------------------------------------- known answers (TESTS/test_hashing.cpp)

GEN_POS = [0.00402186, 0.0388124, 0.0567817, 0.0668489, 0.0915063, 0.12783, 0.198618, 0.199045, 0.250093, 0.259338,
           0.293267, 0.294899, 0.316173, 0.353332, 0.354553, 0.357123, 0.359118, 0.395468, 0.41749, 0.421739,
           0.453347, 0.471302, 0.535031, 0.548733, 0.574022, 0.604538, 0.620419]


def test_cm_between_known_answers():
    """TESTS/test_hashing.cpp:122-157 (wordSize 4; second word inside and beyond the vector)."""
    g = np.array(GEN_POS, np.float32)

    def want(a, b):
        return 100.0 * float(np.float32(g[b] - g[a]))

    gl = [float(x) for x in g]
    assert api.cmBetween(0, 3, gl, 4) == want(0, 15)
    assert api.cmBetween(0, 5, gl, 4) == want(0, 23)
    assert api.cmBetween(3, 5, gl, 4) == want(12, 23)
    assert api.cmBetween(0, 10, gl, 4) == want(0, 26)
    assert api.cmBetween(1, 10, gl, 4) == want(4, 26)


def test_match_known_answers():
    """TESTS/test_hashing.cpp:73-110."""
    m = api.Match(4)
    assert m.getWordSize() == 4 and m.getGaps() == 0 and m.getInterval() == [0, 0]
    m.addGap()
    m.addGap()
    assert m.getGaps() == 2
    m.extend(5)
    assert m.getInterval()[1] == 5
    m = api.Match(4, 7)
    assert m.getWordSize() == 4 and m.getGaps() == 0 and m.getInterval() == [7, 7]
    m.extend(5)
    assert m.getInterval()[1] == 7
    m.extend(8)
    assert m.getInterval()[1] == 8


# ---------------------------------------------------------------- candidate stream

def restate_candidates(alleles, gen, individuals, *, jobs=1, job_ind=1, min_m=1.0, gap=1, skip=0.0, min_maf=0.0,
                       word_size=64, haploid=True, max_seeds=0, read_ahead=10):
    """FastSMC.cpp:118-235 with the emission order this product defines (ascending lower*n+higher per flush).
    ``alleles`` are the raw alleles of EVERY haplotype of the file; ``individuals`` the ones the job loaded.
    ``word_size`` = hashingWordSize, ``haploid``: matches keyed by haplotype or by individual pairs
    (ExtendHash.hpp:47-70), ``max_seeds``/``read_ahead``: large seeds are split by the words read ahead
    (SeedHash.hpp:41-85, FastSMC.cpp:186-195)."""
    import math

    n_tot = alleles.shape[0]
    sample_size = n_tot // 2
    rows = [2 * d + h for d in individuals for h in (0, 1)]
    n = len(rows)
    window = w_i = w_j = 0
    above = False
    if jobs > 1 or job_ind > 1 or True:
        window = math.ceil(math.sqrt((2.0 * sample_size ** 2 - sample_size) * 2.0 / jobs))
        window += window % 2
        w_i, cpt_job, cpt_tot = 1, 1, 1
        while cpt_tot < job_ind:
            w_i += 1
            cpt_job += 2
            cpt_tot += cpt_job
        w_j = math.ceil(np.float32(cpt_job - (cpt_tot - job_ind)) / 2)
        above = (cpt_job - (cpt_tot - job_ind)) % 2 == 1

    def in_job(gi, gj):  # SeedHash.hpp:93-128, gi the later haplotype of the pair
        bi, bj = (w_i - 1) * window, (w_j - 1) * window
        if job_ind == jobs:
            return gi >= bi and gj >= bj and gj < bj + (gi - bi)
        if bi <= gi < bi + window and bj <= gj < bj + window:
            below = gj < bj + (gi - bi)
            return below if above else not below
        return False

    keep = np.arange(alleles.shape[1])
    if min_maf > 0:
        maf = (alleles.sum(axis=0) / float(n_tot)).astype(np.float32)
        keep = keep[~((maf < np.float32(min_maf)) | (maf > np.float32(1) - np.float32(min_maf)))]
    W = word_size
    words = len(keep) // W
    gen = np.asarray(gen, np.float32)
    open_matches = {}
    out = []
    scale = 1 if haploid else 2  # locationToPair: the first haplotype of each individual

    def emit(keys):
        for key in sorted(keys):
            s, e = open_matches.pop(key)
            end_site = min(W * e + W - 1, len(gen) - 1)
            if 100.0 * float(np.float32(gen[end_site] - gen[W * s])) >= float(np.float32(min_m)):
                out.append((scale * (key // n), scale * (key % n), W * s, W * e + W - 1))

    def word_of(local, w):
        return alleles[rows[local], keep[W * w:W * w + W]].tobytes()

    def extend_all_pairs(seeds, w, cur, words_read):  # SeedHash::extendAllPairs (SeedHash.hpp:62-135)
        for members in seeds.values():
            if max_seeds != 0 and len(members) > max_seeds and w + 1 < words_read:
                sub = defaultdict(list)  # SeedHash::subHash: the seed's members by their NEXT word
                for local in members:
                    sub[word_of(local, w + 1)].append(local)
                extend_all_pairs(sub, w + 1, cur, words_read)
                continue
            for x in range(len(members)):
                for y in range(x + 1, len(members)):
                    lo, hi = sorted((members[x], members[y]))
                    if in_job(rows[hi], rows[lo]):
                        a, b = (lo, hi) if haploid else (lo // 2, hi // 2)   # pairToLocation
                        m = open_matches.setdefault(a * n + b, [cur, w])     # extendPair: [CURRENT word, w]
                        m[1] = max(m[1], w)

    # the word buffer (FastSMC.cpp:186-195): read_ahead words before the first is processed, then one more per word
    words_read = cur = 0
    while True:
        while words_read < words:
            words_read += 1
            if words_read >= read_ahead:
                break
        if cur >= words_read:
            break
        w = cur
        seeds = defaultdict(list)
        for local in range(n):
            seeds[word_of(local, w)].append(local)
        if np.float32(len(seeds)) / np.float32(n) > np.float32(skip):
            extend_all_pairs(seeds, w, cur, words_read)
            emit([k for k, m in open_matches.items() if m[1] < w - gap])
        else:
            for m in open_matches.values():
                m[1] = w
        cur += 1
    emit(list(open_matches))
    return out


@pytest.fixture(scope="module")
def hash_files(tmp_path_factory):
    haps = synth.make_haps(80, 1500, seed=11, cm_per_mb=25.0, switch_per_cm=0.5, noise=1e-3)
    root = str(tmp_path_factory.mktemp("hash") / "syn")
    synth.write_haps_files(root, haps)
    return root, haps


def _params(root, **kw):
    p = api.DecodingParams()
    p.inFileRoot = root
    p.FastSMC = True
    p.hashing = True
    p.foldData = True
    p.useKnownSeed = True
    for k, v in kw.items():
        setattr(p, k, v)
    return p


@pytest.mark.parametrize("opts", [dict(), dict(min_m=0.5, gap=0), dict(min_m=2.0, gap=3), dict(skip=0.9),
                                  dict(min_maf=0.05, min_m=0.8), dict(foldData=False)])
def test_candidates_match_restatement(hash_files, opts):
    root, haps = hash_files
    p = _params(root, **opts)
    data = api.Data(p)
    got = [tuple(c) for c in api.hashingCandidates(data, p)]
    kw = {k: v for k, v in opts.items() if k != "foldData"}
    want = restate_candidates(haps.alleles, (haps.cm / 100.0).astype(np.float32), list(range(40)), **kw)
    assert got == want
    if not opts.get("skip"):
        assert len(want) > 20
    assert all(a < b and f % 64 == 0 and t % 64 == 63 and t < 1500 for a, b, f, t in want)


@pytest.mark.parametrize("jobs", [4, 9])
def test_job_windows_partition_the_candidates(hash_files, jobs):
    """Every job sees its own square of the haplotype-pair grid (SeedHash.hpp:93-128): per job the stream equals the
    restatement, and over all jobs each candidate of the single-job run appears exactly once."""
    root, haps = hash_files
    gen = (haps.cm / 100.0).astype(np.float32)
    p1 = _params(root, min_m=0.8)
    whole = {tuple(c) for c in api.hashingCandidates(api.Data(p1), p1)}
    seen = []
    for j in range(1, jobs + 1):
        p = _params(root, min_m=0.8, jobs=jobs, jobInd=j)
        data = api.Data(p)
        individuals = O.job_individuals(40, jobs, j)
        got = [tuple(c) for c in api.hashingCandidates(data, p)]
        assert got == restate_candidates(haps.alleles, gen, individuals, jobs=jobs, job_ind=j, min_m=0.8)
        rows = [2 * d + h for d in individuals for h in (0, 1)]
        seen += [tuple(sorted((rows[a], rows[b]))) + (f, t) for a, b, f, t in got]
    assert len(seen) == len(set(seen))
    assert set(seen) == whole and len(whole) > 20


OTHER_KNOBS = [dict(hashingWordSize=32, min_m=0.5), dict(hashingWordSize=17, min_m=0.3, gap=2),
               dict(haploid=False, min_m=0.8), dict(haploid=False, hashingWordSize=48, gap=0, min_m=0.4),
               dict(max_seeds=3, min_m=0.5), dict(max_seeds=2, constReadAhead=3, min_m=0.3, gap=2),
               dict(max_seeds=1, hashingWordSize=8, min_m=0.2, skip=0.05),
               dict(max_seeds=2, haploid=False, hashingWordSize=16, min_m=0.4, constReadAhead=5)]
_RESTATE_NAMES = {"hashingWordSize": "word_size", "constReadAhead": "read_ahead"}


def restate_kwargs(opts):
    return {_RESTATE_NAMES.get(k, k): v for k, v in opts.items() if k != "foldData"}


@pytest.mark.parametrize("opts", OTHER_KNOBS)
def test_word_size_individual_pairs_and_split_seeds(hash_files, opts):
    """hashingWordSize != 64, haploid = false and max_seeds != 0 against the restatement of the reference's loops."""
    root, haps = hash_files
    p = _params(root, **opts)
    data = api.Data(p)
    got = [tuple(c) for c in api.hashingCandidates(data, p)]
    want = restate_candidates(haps.alleles, (haps.cm / 100.0).astype(np.float32), list(range(40)),
                              **restate_kwargs(opts))
    assert got == want
    assert len(want) > 10
    W = opts.get("hashingWordSize", 64)
    assert all(f % W == 0 and t % W == W - 1 and t < 1500 for _, _, f, t in want)
    if not opts.get("haploid", True):
        assert all(a % 2 == 0 and b % 2 == 0 and a <= b for a, b, _, _ in want)
        assert any(a == b for a, b, _, _ in want) or opts.get("max_seeds")   # the two haplotypes of one individual


def test_split_seeds_change_the_candidates(hash_files):
    """max_seeds really bites on this cohort (else the cases above would prove nothing): the list differs from the
    one without it, and some interval ends beyond the last word its pair was seen to share without look-ahead."""
    root, haps = hash_files
    gen = (haps.cm / 100.0).astype(np.float32)
    plain = restate_candidates(haps.alleles, gen, list(range(40)), min_m=0.5)
    split = restate_candidates(haps.alleles, gen, list(range(40)), min_m=0.5, max_seeds=3)
    assert plain != split and len(split) > 10


def test_hashing_parameters_are_checked(hash_files):
    root, _ = hash_files
    p = _params(root)
    data = api.Data(p)
    for bad in (0, 65):
        p.hashingWordSize = bad
        with pytest.raises(RuntimeError, match="hashingWordSize"):
            api.hashingCandidates(data, p)
    p.hashingWordSize = 64
    p.constReadAhead = 0
    with pytest.raises(RuntimeError, match="constReadAhead"):
        api.hashingCandidates(data, p)


def test_individuals_word_known_answers():
    """TESTS/test_hashing.cpp:33-71 ("individuals"): with words of 8 sites, markers set at (word 0, bit 0), (word 1,
    bit 2) and (word 2, bits 2 and 3) hash to 1, 4 and 12 -- `getWordString` "00000001", "00000100", "00001100" -- and
    an untouched word to 0.  The product has no per-individual ring buffer (the reference's `Individuals` keeps
    `numReadAhead` words and addresses them modulo that; `clear` wipes a slot for re-use): the identification step reads
    whole words from the packed matrix, so what is pinned here is the word PACKER for hashingWordSize != 64 --
    bit b of word w is the allele of site w * wordSize + b (FastSMC.cpp:176-186: setMarker(word, snp_ctr))."""
    word_size, n_words = 8, 10  # ("check up to 10", test_hashing.cpp:41-45)
    S = word_size * n_words + 5  # (+ an incomplete last word: never hashed, FastSMC.cpp:186-195)
    alleles = np.zeros((6, S), np.uint8)
    for word, bit in ((0, 0), (1, 2), (2, 2), (2, 3)):  # ind.setMarker(0, 0); setMarker(1, 2); setMarker(2, 2); setMarker(2, 3)
        alleles[5, word * word_size + bit] = 1          # (idNum 5, test_hashing.cpp:35)
    bp = np.arange(1, S + 1, dtype=np.int64) * 1000
    cm = bp * 1e-6
    data = api.Data.from_arrays(alleles, bp, cm, False, True)  # (no folding: a bit is the file's allele)
    p = api.DecodingParams()
    p.hashingWordSize = word_size
    words = api.hashingWords(data, p)
    assert words.shape == (6, n_words) and words.dtype == np.uint64
    assert [int(x) for x in words[5, :3]] == [1, 4, 12]                      # getWordHash(0 / 1 / 2)
    strings = [format(int(x), f"0{word_size}b") for x in words[5, :3]]     # boost::to_string: highest bit first
    assert strings == ["00000001", "00000100", "00001100"]
    assert not words[5, 3:].any() and not words[:5].any()                     # getWordHash(i) == 0, "00000000"
    # the default word size takes the packed genotype words as they are
    p.hashingWordSize = 64
    w64 = api.hashingWords(data, p)
    assert w64.shape == (6, S // 64)
    assert int(w64[5, 0]) == (1 << 0) | (1 << 10) | (1 << 18) | (1 << 19)
