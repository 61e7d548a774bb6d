"""The .bibd wire format (next-row f2): the reference's own fixture and known answers
(ASMC_SRC/TESTS/test_binary_data_reader.cpp:24-88, data/binary_output.bibd.gz copied to tests/golden/)."""
import os

import pytest

from fastsmc_amd import api

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "binary_output.bibd.gz")


def test_default_line_known_answers():  # test_binary_data_reader.cpp:24-46
    line = api.IbdPairDataLine()
    assert (line.ind1FamId, line.ind1Id, line.ind1Hap) == ("0_00", "0_00", -1)
    assert (line.chromosome, line.ibdStart, line.ibdEnd) == (-1, -1, -1)
    assert line.toString() == "0_00\t0_00\t-1\t0_00\t0_00\t-1\t-1\t-1\t-1\t-1"
    line.lengthInCentimorgans = 1.2
    line.postEst = 2.3
    line.mapEst = 3.4
    assert line.toString() == "0_00\t0_00\t-1\t0_00\t0_00\t-1\t-1\t-1\t-1\t1.2\t-1\t2.3\t3.4"


def test_reference_fixture_known_answers():  # test_binary_data_reader.cpp:48-88
    r = api.BinaryDataReader(GOLDEN)
    l1 = r.getNextLine()
    assert (l1.ind1FamId, l1.ind1Id, l1.ind1Hap, l1.ind2FamId, l1.ind2Id, l1.ind2Hap) == \
        ("1_94", "1_94", 1, "1_104", "1_104", 1)
    assert (l1.chromosome, l1.ibdStart, l1.ibdEnd) == (1, 8740, 1660011)
    assert l1.lengthInCentimorgans == pytest.approx(1.86962, rel=1e-5)
    assert l1.ibdScore == pytest.approx(0.403475, rel=1e-5)
    assert l1.postEst == pytest.approx(146.203, rel=1e-5)
    assert l1.mapEst == pytest.approx(24.9999, rel=1e-5)
    l2 = r.getNextLine()
    assert (l2.ibdStart, l2.ibdEnd) == (1679626, 1679626)
    assert l2.lengthInCentimorgans == pytest.approx(0.0, abs=1e-7)
    assert l2.ibdScore == pytest.approx(0.0175673, rel=1e-5)
    assert l2.postEst == pytest.approx(18029.8, rel=1e-5)
    n = 2
    while r.moreLinesInFile():
        r.getNextLine()
        n += 1
    assert n == 1520
    with pytest.raises(RuntimeError):
        r.getNextLine()
