"""Committed golden vectors (tests/golden/): the oracle must keep reproducing them (CPU), and the HIP path is
compared with the committed numbers as well (GPU)."""
import os

import numpy as np
import pytest

from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "small_problem_oracle.npz"))


def test_oracle_reproduces_committed_vectors():
    import importlib.util

    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    now = mod.build()
    for key in ("records", "posterior_first8", "mean_first8", "map_first8", "pairs"):
        np.testing.assert_array_equal(now[key], GOLD[key], err_msg=key)
    assert float(now["e1_checksum"]) == float(GOLD["e1_checksum"])
    assert GOLD["records"].size > 30


@pytest.mark.gpu
def test_gpu_reproduces_committed_vectors(small_problem):
    from fastsmc_amd import capi

    pm = small_problem["model"]
    ctx = capi.Context(0)
    model = ctx.create_model(pm)
    ctx.upload_haps(small_problem["bits"], pm.S)
    pairs = GOLD["pairs"].view(capi.PAIR_DTYPE).reshape(-1)
    got = ctx.decode_ibd(model, pairs, capi.whole_sequence_groups(pairs.size, pm.S))
    want = GOLD["records"]
    for f_got, f_want in (("pair", "pair"), ("start", "start"), ("end", "end"), ("prob", "prob"),
                          ("post_mean", "postMean"), ("map", "map")):
        np.testing.assert_array_equal(got[f_got], want[f_want], err_msg=f_got)
    ctx.upload_worklist(pairs[:8], capi.whole_sequence_groups(8, pm.S))
    post = ctx.decode_posteriors(model)[0]
    np.testing.assert_array_equal(post[:, :, :8], GOLD["posterior_first8"])
    mean, mp = ctx.decode_per_pair(model, pm.exp_times)
    np.testing.assert_array_equal(mean, GOLD["mean_first8"])
    np.testing.assert_array_equal(mp, GOLD["map_first8"])
    ctx.close()
