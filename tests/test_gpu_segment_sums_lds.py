"""Where the open segments' per-state posterior sums live (HMM.cpp:1212-1229, the input of the segment ages): a launch
smaller than the chip keeps them in LDS, every other launch in the wave's workspace (fsmc_kernels.h, KParams::spsLds;
fsmc_ctx_last_segment_sums_in_lds).  Every record field is the oracle's either way, bit for bit."""
import os

import numpy as np
import pytest

from fastsmc_amd import capi, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu

AGES = capi.FSMC_WANT_MEAN | capi.FSMC_WANT_MAP


def _pairs_array(pairs):
    return np.array(pairs, dtype=np.uint32).view(capi.PAIR_DTYPE).reshape(-1)


def _assert_records_equal(got, want):
    assert got.size == want.size
    for f_got, f_want in (("pair", "pair"), ("start", "start"), ("end", "end"), ("prob", "prob"),
                          ("post_mean", "postMean"), ("map", "map")):
        np.testing.assert_array_equal(got[f_got], want[f_want], err_msg=f_got)


@pytest.fixture
def no_lds_switch():
    yield
    os.environ.pop("FSMC_DIAG_NO_SPS_LDS", None)


@pytest.mark.parametrize("K,time", [(69, 50), (50, 50), (100, 200), (33, 200), (128, 200)])
def test_small_launch_keeps_the_sums_in_lds_and_matches_the_oracle(small_problem, no_lds_switch, K, time):
    """A handful of groups with segment ages: the sums are in LDS (except where a wave's share of LDS would pass a
    workgroup's 64 KiB: the 128-state member); with the diagnostic switch they are in the workspace; the records are
    the oracle's in both runs -- segments that open and close many times per window, ragged groups, sub-windows."""
    sp = small_problem
    if K == 69:
        pm = sp["model"]
    else:
        _, derived, _ = synth.fold_and_pack(sp["haps"].alleles)
        pm = O.prepare_model(synth.make_model_tables(K), sp["gen"], sp["haps"].bp, derived, 64, time=time)
    folded = sp["folded"]
    S = pm.S
    pairs = O.enumerate_all_pairs(32)[:230]
    groups = capi.whole_sequence_groups(len(pairs), S)
    want = O.decode_pairs_ibd(pm, folded, pairs, batch_size=64)
    assert want.size > 20
    ctx = capi.Context(0)
    model = ctx.create_model(pm)
    ctx.upload_haps(sp["bits"], S)
    for switch, in_lds in ((None, K != 128), ("1", False)):
        if switch:
            os.environ["FSMC_DIAG_NO_SPS_LDS"] = switch
        got = ctx.decode_ibd(model, _pairs_array(pairs), groups, flags=AGES)
        assert ctx.last_segment_sums_in_lds() == in_lds
        _assert_records_equal(got, want)
    os.environ.pop("FSMC_DIAG_NO_SPS_LDS", None)
    # without segment ages there are no sums to keep
    ctx.decode_ibd(model, _pairs_array(pairs), groups, flags=0)
    assert not ctx.last_segment_sums_in_lds()
    ctx.close()


def test_a_launch_that_fills_the_chip_keeps_the_sums_in_the_workspace(small_problem):
    """More groups than the chip's LDS has room for beside the kernel's own (four waves a CU at 69 states): the sums stay
    in the workspace; sampled groups against the oracle."""
    sp = small_problem
    pm, folded = sp["model"], sp["folded"]
    S = pm.S
    ctx = capi.Context(0)
    n_cu = ctx.info()["n_cu"]
    model = ctx.create_model(pm)
    ctx.upload_haps(sp["bits"], S)
    ctx.set_pairing(0)  # (half-full groups would otherwise share wavefronts: the paired kernel keeps no sums in LDS)
    allp = O.enumerate_all_pairs(32)
    n_groups = 4 * n_cu + 8  # one pair per group, 80-site windows: many groups, little work
    pairs = [allp[(7 * i) % len(allp)] for i in range(n_groups)]
    groups = np.zeros(n_groups, capi.GROUP_DTYPE)
    for g in range(n_groups):
        frm = (37 * g) % (S - 90)
        groups[g] = (g, 1, frm, frm + 80, frm, frm + 80)
    got = ctx.decode_ibd(model, _pairs_array(pairs), groups, flags=AGES)
    assert not ctx.last_segment_sums_in_lds()
    # ... and one group fewer than the limit of four waves a CU: LDS, the same records for the groups both launches hold
    few = 4 * n_cu - 3
    got_few = ctx.decode_ibd(model, _pairs_array(pairs[:few]), groups[:few], flags=AGES)
    assert ctx.last_segment_sums_in_lds()
    ctx.close()
    key = lambda r: np.lexsort((r["start"], r["pair"]))  # noqa: E731
    a, b = got[got["pair"] < few], got_few
    np.testing.assert_array_equal(a[key(a)], b[key(b)])
    n_checked = 0
    for g in range(0, n_groups, 53):
        frm = int(groups[g]["from"])
        pa, pb = pairs[g]
        ob = (folded[pa] ^ folded[pb])[None, frm:frm + 80]
        hb = (folded[pa] & folded[pb])[None, frm:frm + 80]
        post, _ = O.decode_batch(pm, ob, hb, frm, frm + 80)
        full = np.zeros((S, pm.K, 1), np.float32)
        full[frm:frm + 80] = post[frm:frm + 80]
        want = O.ibd_scan_pair(pm, full, 0, frm, frm + 80, want_mean=True, want_map=True, pair_ordinal=g)
        mine = got[got["pair"] == g]
        mine = mine[np.argsort(mine["start"])]
        assert mine.size == want.size
        for f_got, f_want in (("start", "start"), ("end", "end"), ("prob", "prob"), ("post_mean", "postMean"), ("map", "map")):
            np.testing.assert_array_equal(mine[f_got], want[f_want], err_msg=f_got)
        n_checked += want.size
    assert n_checked > 0
