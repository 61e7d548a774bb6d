"""The reference holds TWO implementations of the decode chain: the batched one the product restates
(HMM::decodeBatch, HMM.cpp:639-1041) and a scalar one "for debugging and pedagogical reasons" (HMM::decode,
HMM.cpp:1460-1721: plain loops over the states, one pair, a division per posterior entry).  Both are restated in
oracle/hmm_oracle.c (fo_decode_batch / fo_decode_scalar); with the model blobs of the reference's golden outputs
missing, their agreement is the closest thing to a reference-held cross-check of the chain: two statements written
from two different pieces of reference code must give the same posterior to float accuracy (1e-5 relative -- the
north star's own tolerance for float posteriors), in array mode (the product's default and every BASELINE
configuration), on sub-windows and for every lane of a batch.  They differ in rounding only (reciprocal-multiply against division; the first site's emission), so they
are NOT bit-identical, and a non-zero difference shows that the comparison is not vacuous."""
import numpy as np

from oracle import oracle as O


def _compare(m, folded, pairs, frm, to):
    ob = np.stack([folded[a] ^ folded[b] for a, b in pairs])
    hb = np.stack([folded[a] & folded[b] for a, b in pairs])
    post, _ = O.decode_batch(m, ob[:, frm:to], hb[:, frm:to], frm, to)  # [S][K][B]
    worst, differs = 0.0, False
    for v in range(len(pairs)):
        scalar = O.decode_scalar(m, ob[v], hb[v], frm, to)  # [K][S]
        batched = post[frm:to, :, v].T
        got = scalar[:, frm:to]
        np.testing.assert_allclose(got.sum(axis=0), 1.0, rtol=0, atol=2e-6)
        # 1e-5 relative on every posterior entry that carries weight (entries below 1e-12 hold no information at fp32)
        np.testing.assert_allclose(got, batched, rtol=1e-5, atol=1e-12)
        differs |= not np.array_equal(got, batched)
        worst = max(worst, float(np.max(np.abs(got - batched) / np.maximum(batched, 1e-12))))
        assert not scalar[:, :frm].any() and not scalar[:, to:].any()
    return worst, differs


def test_scalar_and_batched_restatements_agree_array_mode(small_problem):
    m, folded = small_problem["model"], small_problem["folded"]
    pairs = [(0, 1), (3, 10), (5, 62), (20, 21), (7, 7 + 32), (40, 41), (2, 63), (11, 12)]
    differs = False
    for frm, to in ((0, m.S), (100, 400), (637, 640), (0, 2)):
        worst, d = _compare(m, folded, pairs, frm, to)
        differs |= d
        assert worst <= 1e-5
    assert differs  # two different roundings of the same mathematics, not one function called twice


def test_sequence_mode_is_where_the_references_two_paths_part(seq_problem):
    """In sequence mode the reference's two implementations are NOT the same function: the batched path's
    `previousAlpha = nextAlpha` / `lastComputedBeta = previousBeta` are Eigen::Map assignments that copy the
    un-normalised half-step result over the neighbouring site's stored vector (HMM.cpp:767, 922), so its posterior of
    site p is built from alpha after the half-step towards p+1 and beta after the half-step towards p-1; the scalar
    path stores the site-step vectors (HMM.cpp:1583-1600, 1669-1686).  The product's contract is the batched path
    (DESIGN.md 3.6); this test records that the two differ there -- both are proper posteriors (columns sum to one), of
    different vectors -- so the scalar statement cross-checks the chain in array mode only."""
    m, folded = seq_problem["model"], seq_problem["folded"]
    a, b = 3, 10
    ob, hb = folded[a] ^ folded[b], folded[a] & folded[b]
    post, _ = O.decode_batch(m, ob[None, :], hb[None, :], 0, m.S)
    scalar = O.decode_scalar(m, ob, hb, 0, m.S)
    np.testing.assert_allclose(scalar.sum(axis=0), 1.0, rtol=0, atol=2e-6)
    np.testing.assert_allclose(post[:, :, 0].sum(axis=1), 1.0, rtol=0, atol=2e-6)
    assert np.max(np.abs(scalar.T - post[:, :, 0])) > 1e-3


def test_scalar_restatement_against_dense_float64(small_problem):
    """... and the scalar statement against the independent float64 forward-backward on the dense K x K matrix
    (tests/test_oracle_dense.py): the same bar the batched statement is held to."""
    from test_oracle_dense import dense_posterior

    m, folded = small_problem["model"], small_problem["folded"]
    for a, b in ((0, 1), (5, 62)):
        ob, hb = folded[a] ^ folded[b], folded[a] & folded[b]
        scalar = O.decode_scalar(m, ob, hb, 100, 400)[:, 100:400].T
        dense = dense_posterior(m, ob[100:400], hb[100:400], 100, 400)
        np.testing.assert_allclose(scalar, dense, rtol=2e-4, atol=1e-7)
