"""Structural check of the oracle against an independent statement of the same HMM: build the dense K x K
transition matrix from (D, B, U, rowRatios, columnRatios) (Transition.java:152-209) and run a
textbook float64 forward-backward.  The O(K) recurrences of the oracle must give the same posterior."""
import numpy as np

from oracle import oracle as O


def dense_T(m, row):
    K = m.K
    D, B, U, RR, cR = (x.astype(np.float64) for x in (m.D[row], m.B[row], m.U[row], m.RR[row], m.col_ratios))
    T = np.zeros((K, K))
    for i in range(K):
        T[i, i] = D[i]
        T[i, :i] = B[:i]
    for i in range(K - 2, -1, -1):
        T[i, i + 1] = U[i]
        for j in range(i + 2, K):
            T[i, j] = T[i, j - 1] * cR[j - 1]
    # the row-ratio form must describe the same matrix
    for i in range(K - 2):
        np.testing.assert_allclose(T[i, i + 2:], RR[i] * T[i + 1, i + 2:], rtol=2e-5, atol=1e-30)
    return T


def emission(m, pos, x, a):
    z, t = (0.0 if x else 1.0), (1.0 if a else 0.0)
    return m.e1[pos].astype(np.float64) + m.e0m1[pos].astype(np.float64) * z + m.e2m0[pos].astype(np.float64) * t


def dense_posterior(m, xbits, abits, frm, to):
    K = m.K
    n = to - frm
    al = np.zeros((n, K))
    be = np.zeros((n, K))
    a = m.pi.astype(np.float64) * emission(m, frm, xbits[0], abits[0])
    al[0] = a / a.sum()
    Ts = {}
    for p in range(frm + 1, to):
        T = Ts.setdefault(int(m.step_row[p]), dense_T(m, int(m.step_row[p])))
        a = emission(m, p, xbits[p - frm], abits[p - frm]) * (al[p - frm - 1] @ T)
        al[p - frm] = a / a.sum()
    be[n - 1] = 1.0 / K
    for p in range(to - 2, frm - 1, -1):
        T = Ts[int(m.step_row[p + 1])]
        b = T @ (emission(m, p + 1, xbits[p + 1 - frm], abits[p + 1 - frm]) * be[p + 1 - frm])
        be[p - frm] = b / b.sum()
    post = al * be
    return post / post.sum(axis=1, keepdims=True)


def test_oracle_matches_dense_float64(small_problem):
    m = small_problem["model"]
    folded = small_problem["folded"]
    pairs = [(0, 1), (3, 10), (5, 62), (20, 21)]
    for frm, to in ((0, m.S), (100, 400)):
        ob = np.stack([(folded[a] ^ folded[b])[frm:to] for a, b in pairs])
        hb = np.stack([(folded[a] & folded[b])[frm:to] for a, b in pairs])
        post, _ = O.decode_batch(m, ob, hb, frm, to)
        for v in range(len(pairs)):
            ref = dense_posterior(m, ob[v], hb[v], frm, to)
            got = post[frm:to, :, v].astype(np.float64)
            np.testing.assert_allclose(got.sum(axis=1), 1.0, rtol=1e-5)
            assert np.max(np.abs(got - ref)) < 2e-5
            big = ref > 1e-3
            assert np.max(np.abs(got[big] / ref[big] - 1.0)) < 1e-3


def test_lanes_are_independent(small_problem):
    """A pair's result must not depend on batch composition or batch size (lane = pair)."""
    m = small_problem["model"]
    folded = small_problem["folded"]
    pairs = [(0, 1), (3, 10), (5, 62), (20, 21), (7, 9), (11, 40), (2, 33), (8, 50)]
    ob = np.stack([folded[a] ^ folded[b] for a, b in pairs])
    hb = np.stack([folded[a] & folded[b] for a, b in pairs])
    post8, _ = O.decode_batch(m, ob, hb, 0, m.S)
    post4, _ = O.decode_batch(m, ob[4:], hb[4:], 0, m.S)
    np.testing.assert_array_equal(post8[:, :, 4:], post4)


def dense_posterior_sequence(m, xbits, abits, frm, to):
    """Sequence mode as the reference's buffers end up (HMM.cpp:760-770, 915-925 and hmm_oracle.h): the stored
    alpha of site p < to-1 is the un-scaled vector after the homozygous half-step towards p+1, the stored beta of
    site p > from the one after the half-step towards p-1."""
    K = m.K
    n = to - frm
    Ts = {}

    def T(row):
        return Ts.setdefault(int(row), dense_T(m, int(row)))

    hom = m.hom.astype(np.float64)
    al = np.zeros((n, K))
    be = np.zeros((n, K))
    a = m.pi.astype(np.float64) * emission(m, frm, xbits[0], abits[0])
    a /= a.sum()
    for p in range(frm + 1, to):
        half = hom[p] * (a @ T(m.gap_row_f[p]))
        al[p - 1 - frm] = half
        a = emission(m, p, xbits[p - frm], abits[p - frm]) * (half @ T(m.site_row_f[p]))
        a /= a.sum()
    al[n - 1] = a
    b = np.full(K, 1.0 / K)
    for p in range(to - 2, frm - 1, -1):
        q = p + 1
        half = T(m.gap_row_b[q]) @ (hom[q] * b)
        be[q - frm] = half
        b = T(m.site_row_b[q]) @ (emission(m, q, xbits[q - frm], abits[q - frm]) * half)
        b /= b.sum()
    be[0] = b
    post = al * be
    return post / post.sum(axis=1, keepdims=True)


def test_oracle_sequence_mode_matches_dense_float64(seq_problem):
    m = seq_problem["model"]
    assert m.sequence and len(np.unique(m.gap_row_f)) > 10
    folded = seq_problem["folded"]
    pairs = [(0, 1), (3, 10), (5, 30), (20, 21)]
    for frm, to in ((0, m.S), (37, 211), (5, 6)):
        ob = np.stack([(folded[a] ^ folded[b])[frm:to] for a, b in pairs])
        hb = np.stack([(folded[a] & folded[b])[frm:to] for a, b in pairs])
        post, _ = O.decode_batch(m, ob, hb, frm, to)
        for v in range(len(pairs)):
            ref = dense_posterior_sequence(m, ob[v], hb[v], frm, to)
            got = post[frm:to, :, v].astype(np.float64)
            np.testing.assert_allclose(got.sum(axis=1), 1.0, rtol=1e-5)
            assert np.max(np.abs(got - ref)) < 2e-5
            big = ref > 1e-3
            assert np.max(np.abs(got[big] / ref[big] - 1.0)) < 1e-3
