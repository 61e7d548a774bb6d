"""The host readers and the pair enumeration on the REFERENCE's own data (tests/golden/fastsmc_example.* are copies of
FILES/FASTSMC_EXAMPLE/example.{hap.gz,samples}; the two *.ibd.gz are the reference's golden outputs for that data,
TESTS/test_fastsmc_regression.cpp:28-161).

The decoding quantities and the genetic map the goldens were made with are missing from the reference checkout, so
their float columns cannot be reproduced -- but everything in them that does not depend on the model is a known
answer for this build's host side, with no GPU:
  * which individuals a job loads (Data.cpp:62-80) -- every id in the golden of job 7 of 9 is one of them;
  * which pairs a job decodes and in what order (HMM::decodeAll, HMM.cpp:283-381; record ids / haplotype numbers of
    HMM::writePairIBD, HMM.cpp:1116-1144) -- the golden's pairs, in file order, are a subsequence of the enumeration;
  * physical positions come from the .hap file (column 3), start <= end.
The readers themselves (Data.cpp:212-248 samples, 397-521 haps, folding 462-509) are checked against a plain parse.
"""
import gzip
import os
import shutil

import numpy as np
import pytest

from fastsmc_amd import api, synth

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def _plain_parse():
    bp, rows = [], []
    with gzip.open(os.path.join(GOLD, "fastsmc_example.hap.gz"), "rt") as f:
        for line in f:
            t = line.split()
            bp.append(int(t[2]))
            rows.append(np.array(t[5:], dtype=np.uint8))
    alleles = np.stack(rows, axis=1)  # [haplotype][site]
    ids = [ln.split()[:2] for ln in open(os.path.join(GOLD, "fastsmc_example.samples")).read().splitlines()[2:]]
    return alleles, np.array(bp, np.int64), ids


@pytest.fixture(scope="module")
def example_root(tmp_path_factory):
    """The example's files under one root, with a synthetic 1 cM/Mb map in the FastSMC format (Data.cpp:98-141)."""
    d = tmp_path_factory.mktemp("example")
    root = str(d / "example")
    shutil.copy(os.path.join(GOLD, "fastsmc_example.hap.gz"), root + ".hap.gz")
    shutil.copy(os.path.join(GOLD, "fastsmc_example.samples"), root + ".samples")
    _, bp, _ = _plain_parse()
    with open(root + ".map", "w") as f:
        for p in bp:
            f.write(f"{int(p)}\t1.0\t{float(p) * 1e-6!r}\n")
    return root


def _params(root, **kw):
    p = api.DecodingParams()
    p.inFileRoot = root
    p.decodingModeString = "array"
    p.foldData = True
    p.usingCSFS = True
    p.batchSize = 32
    p.hashing = False
    p.FastSMC = True
    p.time = 50
    p.noConditionalAgeEstimates = True
    p.doPerPairMAP = True
    p.doPerPairPosteriorMean = True
    p.outputIbdSegmentLength = True
    p.useKnownSeed = True
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def test_readers_on_the_reference_example(example_root):
    alleles, bp, ids = _plain_parse()
    assert alleles.shape == (300, 6760) and len(ids) == 150
    data = api.Data(_params(example_root))
    assert data.sites == 6760 and data.sampleSize == 150 and data.haploidSampleSize == 300
    assert list(data.FamIDList) == [a for a, _ in ids] and list(data.IIDList) == [b for _, b in ids]
    np.testing.assert_array_equal(np.array(data.physicalPositions, np.int64), bp)
    np.testing.assert_array_equal(np.array(data.geneticPositions, np.float32),
                                  (bp.astype(np.float64) * 1e-6 / 100.0).astype(np.float32))
    bits, derived, flipped = synth.fold_and_pack(alleles)  # minor-allele folding over ALL samples (Data.cpp:462-509)
    np.testing.assert_array_equal(data.packed_bits(), bits)
    np.testing.assert_array_equal(np.array(data.siteWasFlippedDuringFolding, bool), flipped)
    np.testing.assert_array_equal(np.array(data.derivedAlleleCounts), derived)
    assert api.Data.countHapLines(example_root) == 6760


def _golden_records(name):
    out = []
    for line in gzip.open(os.path.join(GOLD, name), "rt"):
        t = line.split("\t")
        out.append((t[1], int(t[2]), t[4], int(t[5]), int(t[6]), int(t[7]), int(t[8])))
    return out


def test_job_7_of_9_enumeration_against_the_reference_golden(example_root):
    recs = _golden_records("fastsmc_example_regression_output_no_hashing.ibd.gz")
    assert len(recs) == 2986  # expectedNumLines, test_fastsmc_regression.cpp:127
    _, bp, _ = _plain_parse()
    p = _params(example_root, jobs=9, jobInd=7)
    data = api.Data(p)
    hmm = api.HMM(data, api.decoding_quantities_from_tables(synth.make_model_tables(69)), p)
    iids = list(data.IIDList)
    # (1) the individuals the job loads
    assert {r[0] for r in recs} | {r[2] for r in recs} <= set(iids)
    # (2) the pairs the job decodes, in order: (IID, hap) of both sides as writePairIBD prints them
    mine = [(iids[a // 2], a % 2 + 1, iids[b // 2], b % 2 + 1) for a, b in hmm.pairsOfJob(9, 7)]
    assert len(mine) == len(set(mine))
    seen = []
    for r in recs:
        key = r[:4]
        if not seen or seen[-1] != key:
            seen.append(key)
    assert len(seen) == len(set(seen)) == 1478  # a pair's segments are contiguous in the file
    it = iter(mine)
    assert all(any(k == m for m in it) for k in seen), "the golden's pairs are not a subsequence of the enumeration"
    # (3) coordinates
    pos = set(int(x) for x in bp)
    assert all(r[4] == 1 and r[5] in pos and r[6] in pos and r[5] <= r[6] for r in recs)


def test_hashing_golden_names_pairs_of_the_whole_cohort(example_root):
    recs = _golden_records("fastsmc_example_regression_output.ibd.gz")
    _, bp, ids = _plain_parse()
    iids = {b for _, b in ids}
    pos = set(int(x) for x in bp)
    assert len(recs) == 1524
    assert all(r[0] in iids and r[2] in iids and r[1] in (1, 2) and r[3] in (1, 2) for r in recs)
    assert all(r[5] in pos and r[6] in pos and r[5] <= r[6] for r in recs)


@pytest.mark.skipif(not os.path.isdir("/root/reference/FILES/EXAMPLE"), reason="reference checkout not present")
def test_asmc_format_readers_on_the_reference_array_example():
    """ASMC-format inputs (plink 4-column .map.gz, Data.cpp:162-210; haps without the FastSMC job logic) on
    FILES/EXAMPLE/exampleFile.n300.array.* -- the data of TESTS/test_HMM.cpp / test_ASMC.cpp.  Build container only."""
    root = "/root/reference/FILES/EXAMPLE/exampleFile.n300.array"
    p = api.DecodingParams()
    p.inFileRoot = root
    p.decodingModeString = "array"
    p.foldData = True
    p.usingCSFS = True
    data = api.Data(p)
    bp, cm, rows = [], [], []
    for line in gzip.open(root + ".hap.gz", "rt"):
        t = line.split()
        bp.append(int(t[2]))
        rows.append(np.array(t[5:], dtype=np.uint8))
    for line in gzip.open(root + ".map.gz", "rt"):
        cm.append(float(line.split()[2]))
    alleles = np.stack(rows, axis=1)
    assert data.sites == len(bp) == len(cm) and data.sampleSize == alleles.shape[0] // 2 == 150
    assert len(data.individuals) > 20  # test_HMM.cpp:33
    np.testing.assert_array_equal(np.array(data.physicalPositions, np.int64), np.array(bp))
    np.testing.assert_allclose(np.array(data.geneticPositions, np.float64), np.array(cm) / 100.0, rtol=1e-6)
    bits, _, _ = synth.fold_and_pack(alleles)
    np.testing.assert_array_equal(data.packed_bits(), bits)
