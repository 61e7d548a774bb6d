"""The one-pass, multi-threaded haps reader (host/data.cpp: Data::readHapsFastSMC; reference Data.cpp:397-565) against a
plain parse of the same file: ragged shapes (haplotypes and sites that fill no whole 64-bit word), a file cut into many
blocks, job windows that load two ranges of individuals, a last line without a newline, the reference's error messages --
and the emission preparation's shuffles (Data.cpp:144-160, 567-599), whose std::rand() seeds are drawn in order while
the shuffles run in parallel, against the oracle's restatement."""
import gzip

import numpy as np
import pytest

from fastsmc_amd import api, synth
from oracle import oracle as O


def _write(root, alleles, bp, cm, newline_at_end=True, gz=True):
    n_hap, S = alleles.shape
    lines = [f"1:{int(bp[s])}_A_G SNP{s} {int(bp[s])} A G " + " ".join("1" if a else "0" for a in alleles[:, s])
             for s in range(S)]
    text = "\n".join(lines) + ("\n" if newline_at_end else "")
    if gz:
        with gzip.open(root + ".hap.gz", "wt") as f:
            f.write(text)
    else:
        open(root + ".hap", "w").write(text)
    with open(root + ".samples", "w") as f:
        f.write("ID_1 ID_2 missing\n0 0 0\n")
        for i in range(n_hap // 2):
            f.write(f"fam{i} ind{i} 0\n")
    with open(root + ".map", "w") as f:
        for s in range(S):
            f.write(f"{int(bp[s])}\t0.0\t{float(cm[s])!r}\n")


def _params(root, tmp_path, jobs=1, job=1):
    p = api.DecodingParams(in_dir=root, decoding_quants=root + ".dq.missing", out_dir=str(tmp_path / "out"), FastSMC=True)
    p.hashing = False
    p.useKnownSeed = True
    p.jobs, p.jobInd = jobs, job
    return p


def _problem(n_ind, S, seed):
    rng = np.random.default_rng(seed)
    freq = rng.uniform(0.02, 0.98, S)
    alleles = (rng.uniform(size=(2 * n_ind, S)) < freq[None, :]).astype(np.uint8)
    bp = np.cumsum(rng.integers(1, 500, S)).astype(np.int64)
    cm = np.cumsum(rng.uniform(1e-4, 1e-2, S))
    return alleles, bp, cm


@pytest.mark.parametrize("n_ind,S,block,newline", [(37, 1003, 0, True), (37, 1003, 4096, False), (70, 130, 700, True),
                                                   (32, 64, 0, True)])
def test_one_pass_reader_equals_a_plain_parse(tmp_path, monkeypatch, n_ind, S, block, newline):
    alleles, bp, cm = _problem(n_ind, S, 5 + S)
    root = str(tmp_path / "x")
    _write(root, alleles, bp, cm, newline_at_end=newline, gz=(S != 130))
    if block:
        monkeypatch.setenv("FSMC_HOST_BLOCK_BYTES", str(block))  # many blocks, each a few lines
    monkeypatch.setenv("FSMC_HOST_THREADS", "5")
    data = api.Data(_params(root, tmp_path))
    bits, derived, flipped = synth.fold_and_pack(alleles)
    assert data.sites == S and data.sampleSize == n_ind and data.chrNumber == 1
    np.testing.assert_array_equal(data.packed_bits(), bits)
    np.testing.assert_array_equal(np.array(data.derivedAlleleCounts), derived)
    np.testing.assert_array_equal(np.array(data.siteWasFlippedDuringFolding), flipped)
    np.testing.assert_array_equal(np.array(data.physicalPositions), bp)
    np.testing.assert_array_equal(np.array(data.geneticPositions, np.float32), (cm / np.float32(100.0)).astype(np.float32))
    # the same file with one thread and one block
    monkeypatch.setenv("FSMC_HOST_THREADS", "1")
    monkeypatch.delenv("FSMC_HOST_BLOCK_BYTES", raising=False)
    again = api.Data(_params(root, tmp_path))
    np.testing.assert_array_equal(again.packed_bits(), data.packed_bits())


def test_job_windows_load_ranges_of_individuals(tmp_path, monkeypatch):
    alleles, bp, cm = _problem(41, 300, 9)
    root = str(tmp_path / "j")
    _write(root, alleles, bp, cm)
    monkeypatch.setenv("FSMC_HOST_BLOCK_BYTES", "2000")
    whole_counts = None
    for job in range(1, 10):  # jobs = 9: diagonal and off-diagonal windows, the last job takes the remainder
        data = api.Data(_params(root, tmp_path, jobs=9, job=job))
        ids = np.array(data.globalIndIndex)
        rows = np.stack([2 * ids, 2 * ids + 1], axis=1).reshape(-1)
        cnt = alleles.sum(axis=0)
        minor_is_one = cnt <= alleles.shape[0] - cnt  # folding looks at the WHOLE file (Data.cpp:465-471)
        folded = np.where(minor_is_one[None, :], alleles, 1 - alleles).astype(np.uint8)
        np.testing.assert_array_equal(data.packed_bits(), synth.pack_bits(folded[rows]))
        counts = np.array(data.derivedAlleleCounts)
        whole_counts = counts if whole_counts is None else whole_counts
        np.testing.assert_array_equal(counts, whole_counts)


def test_reader_errors_are_the_reference_messages(tmp_path):
    alleles, bp, cm = _problem(8, 20, 3)
    root = str(tmp_path / "e")
    bad_bp = bp.copy()
    bad_bp[7] = bad_bp[6]
    _write(root, alleles, bad_bp, cm)
    with pytest.raises(RuntimeError, match="ordered by increasing physical position"):
        api.Data(_params(root, tmp_path))
    _write(root, alleles, bp, cm)
    text = gzip.open(root + ".hap.gz", "rt").read().replace(" 0 1", " 0 2", 1)
    gzip.open(root + ".hap.gz", "wt").write(text)
    with pytest.raises(RuntimeError, match="not '0' or '1'"):
        api.Data(_params(root, tmp_path))
    gzip.open(root + ".hap.gz", "wt").write(text.replace(" 0 2", " 0", 1))
    with pytest.raises(RuntimeError, match="wrong length"):
        api.Data(_params(root, tmp_path))


def test_parallel_shuffles_keep_the_seed_order(monkeypatch):
    """calculateUndistinguishedCounts: the same counts whatever the number of threads, and those of the oracle's
    restatement of Data.cpp:144-160, 567-599 (libstdc++ shuffle seeded from glibc rand, srand(1234))."""
    rng = np.random.default_rng(17)
    n_hap, S = 120, 700
    alleles = (rng.uniform(size=(n_hap, S)) < rng.uniform(0.0, 1.0, S)[None, :]).astype(np.uint8)
    alleles[:, 5] = 0  # a monomorphic site: draws are skipped where the reference skips them
    alleles[:, 6] = 1
    bp = np.arange(1, S + 1, dtype=np.int64) * 100
    cm = np.arange(S) * 1e-3
    _, derived, _ = synth.fold_and_pack(alleles)
    want = O.undistinguished_counts(derived, np.full(S, n_hap, np.int32), 50, fold=True)
    for threads in ("1", "7"):
        monkeypatch.setenv("FSMC_HOST_THREADS", threads)
        data = api.Data.from_arrays(alleles, bp, cm, True, True)
        got = np.array(data.calculateUndistinguishedCounts(50))
        np.testing.assert_array_equal(got, want)
