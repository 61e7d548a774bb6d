"""The -O3 -mavx2 build of the oracle (timed by bench.py's cpu_baseline leg) against the -O2 checker build: identical
IBD records and posteriors, serial and with one batch per thread."""
import numpy as np

from oracle import oracle as O


def test_avx2_build_and_threads_are_bit_identical(small_problem):
    pm = small_problem["model"]
    pairs = O.enumerate_all_pairs(32)[:150]
    want = O.decode_pairs_ibd(pm, small_problem["folded"], pairs, batch_size=32)
    O.select_build("avx2")
    try:
        got = O.decode_pairs_ibd(pm, small_problem["folded"], pairs, batch_size=32)
        got_mt = O.decode_pairs_ibd(pm, small_problem["folded"], pairs, batch_size=32, threads=4)
        folded = small_problem["folded"]
        ob = np.stack([folded[a] ^ folded[b] for a, b in pairs[:8]])
        hb = np.stack([folded[a] & folded[b] for a, b in pairs[:8]])
        post_fast, _ = O.decode_batch(pm, ob, hb, 0, pm.S)
    finally:
        O.select_build("ref")
    post_ref, _ = O.decode_batch(pm, ob, hb, 0, pm.S)
    assert want.size > 10
    assert want.tobytes() == got.tobytes() == got_mt.tobytes()
    np.testing.assert_array_equal(post_fast, post_ref)
