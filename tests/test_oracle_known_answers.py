"""The reference's own known-answer tests for the helpers on the decode path, restated against the
oracle (reference: ASMC_SRC/TESTS/test_hmm_utils.cpp).  This is what pins the oracle's helpers."""
import numpy as np
import pytest

from oracle import oracle as O


def test_round_morgans_known_answers():  # test_hmm_utils.cpp:180-202
    for prec in (0, 1, 2):
        assert O.round_morgans(0.4, prec, 0.5) == np.float32(0.5)
    assert O.round_morgans(0.0, 5, 0.5) == np.float32(0.5)
    assert O.round_morgans(-1.0, 7, 0.5) == np.float32(0.5)
    a = 0.123456
    for prec, want in ((0, 0.1), (1, 0.12), (2, 0.123), (3, 0.1235), (4, 0.12346)):
        assert O.round_morgans(a, prec, 1e-10) == np.float32(want)


def test_round_physical_known_answers():  # test_hmm_utils.cpp:204-229
    for v in (-1, 0, 1):
        for prec in (0, 1, 2):
            assert O.round_physical(v, prec) == 1
    for prec, want in enumerate((100000, 120000, 123000, 123500, 123460, 123456)):
        assert O.round_physical(123456, prec) == want


@pytest.mark.parametrize("vecx", [1, 4, 8, 16])
def test_scaling_batch_known_answers(vecx):  # test_hmm_utils.cpp:231-296
    batch, states = max(2, 2 * vecx), 2
    data = np.tile(1.0 + np.arange(batch, dtype=np.float32), states)
    scal, sums = O.calculate_scaling_batch(data, batch, states)
    want = np.float32(1.0) / (2 * (1.0 + np.arange(batch, dtype=np.float32)))
    np.testing.assert_allclose(scal, want, rtol=1e-7)
    scalings = np.arange(batch, dtype=np.float32) + 5.0
    out = O.apply_scaling_batch(data, scalings, batch, states)
    np.testing.assert_array_equal(out, np.tile((1.0 + np.arange(batch, dtype=np.float32)) * scalings, states))


def test_window_padding_known_answers():  # test_hmm_utils.cpp:298-332
    g = np.array([0.12, 0.23, 0.34, 0.45, 0.56, 0.67], np.float32)
    assert [O.get_from_position(g, 4, c) for c in (1, 21, 23, 30, 45, 60)] == [3, 2, 1, 1, 0, 0]
    assert [O.get_from_position(g, 0, c) for c in (1e-6, 1, 10)] == [0, 0, 0]
    assert [O.get_to_position(g, 1, c) for c in (1, 10, 12, 30, 40, 60)] == [3, 3, 4, 5, 6, 6]
    assert [O.get_to_position(g, 6, c) for c in (1e-6, 1, 10)] == [6, 6, 6]


def test_subset_xor_and_known_answers():  # test_hmm_utils.cpp:32-56
    v1 = [0, 0, 1, 1, 0, 0]
    v2 = [0, 0, 1, 1, 1, 0]
    assert O.subset_xor(v1, v2).tolist() == [0, 0, 0, 0, 1, 0]
    assert O.subset_xor(v1, v2, 0, 6).tolist() == [0, 0, 0, 0, 1, 0]
    assert O.subset_xor(v1, v2, 3, 5).tolist() == [0, 1]
    assert O.subset_and(v1, v2).tolist() == [0, 0, 1, 1, 0, 0]
    assert O.subset_and(v1, v2, 3, 5).tolist() == [1, 0]
