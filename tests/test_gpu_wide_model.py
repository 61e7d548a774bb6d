"""The wide-model kernel (K = 256: lane = pair, four waves per group; fastsmc_amd/csrc/fsmc_kernels_w2.h) against the
oracle, bit for bit: state thresholds inside the first quarter of the states, across two and across three quarters (the
scan's sum then crosses waves), ragged groups, sub-windows with scan windows inside them, the checkpoint/recompute
layout, with and without segment ages, and the posterior dump."""
import numpy as np
import pytest

from fastsmc_amd import capi, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu

FIELDS = (("pair", "pair"), ("start", "start"), ("end", "end"), ("prob", "prob"), ("post_mean", "postMean"),
          ("map", "map"))


@pytest.fixture(scope="module")
def wide():
    tables = synth.make_model_tables(256)
    haps = synth.make_haps(64, 260, seed=5, cm_per_mb=25.0, switch_per_cm=0.6)
    bits, derived, flipped = synth.fold_and_pack(haps.alleles)
    folded = np.where(flipped[None, :], 1 - haps.alleles, haps.alleles).astype(np.uint8)
    gen = (haps.cm / 100.0).astype(np.float32)
    return dict(tables=tables, haps=haps, bits=bits, derived=derived, folded=folded, gen=gen)


def _model(w, time):
    return O.prepare_model(w["tables"], w["gen"], w["haps"].bp, w["derived"], 64, time=time)


def _pairs_array(pairs):
    return np.array(pairs, dtype=np.uint32).view(capi.PAIR_DTYPE).reshape(-1)


def _assert_records_equal(got, want):
    assert got.size == want.size
    for f_got, f_want in FIELDS:
        np.testing.assert_array_equal(got[f_got], want[f_want], err_msg=f_got)


@pytest.mark.parametrize("time,flags", [(200, capi.FSMC_WANT_MEAN | capi.FSMC_WANT_MAP), (5000, capi.FSMC_WANT_MAP),
                                        (20000, capi.FSMC_WANT_MEAN | capi.FSMC_WANT_MAP), (20000, 0)])
@pytest.mark.parametrize("limit", [0, 24 << 20])
def test_records_identical_to_oracle(wide, time, flags, limit):
    pm = _model(wide, time)
    assert pm.K == 256
    pairs = O.enumerate_all_pairs(32)[:151]  # 64 + 64 + 23: the last group has a ragged second and two empty quarters
    want = O.decode_pairs_ibd(pm, wide["folded"], pairs, batch_size=64, want_mean=bool(flags & capi.FSMC_WANT_MEAN),
                              want_map=bool(flags & capi.FSMC_WANT_MAP))
    ctx = capi.Context(0)
    if limit:
        ctx.set_workspace_limit(limit)
    model = ctx.create_model(pm)
    ctx.upload_haps(wide["bits"], pm.S)
    got = ctx.decode_ibd(model, _pairs_array(pairs), capi.whole_sequence_groups(len(pairs), pm.S), flags)
    if limit:
        assert ctx.info()["max_chunks"] > 1
    ctx.close()
    assert want.size > 10
    _assert_records_equal(got, want)


def test_windows_and_posterior_dump(wide):
    pm = _model(wide, 5000)
    allp = O.enumerate_all_pairs(32)
    wins = [(0, 64, 0, 260, 0, 260), (64, 17, 3, 255, 10, 250), (81, 5, 100, 101, 100, 101), (86, 33, 200, 203, 201, 203),
            (119, 64, 31, 188, 31, 188)]
    n = wins[-1][0] + wins[-1][1]
    pairs = allp[700:700 + n]
    groups = np.zeros(len(wins), capi.GROUP_DTYPE)
    for g, w in zip(groups, wins):
        g["first_pair"], g["n_pairs"], g["from"], g["to"], g["scan_from"], g["scan_to"] = w
    ctx = capi.Context(0)
    model = ctx.create_model(pm)
    ctx.upload_haps(wide["bits"], pm.S)
    got = ctx.decode_ibd(model, _pairs_array(pairs), groups)
    post = ctx.decode_posteriors(model)
    ctx.close()
    want = []
    folded = wide["folded"]
    for gi, (first, cnt, frm, to, sfrm, sto) in enumerate(wins):
        sub = pairs[first:first + cnt]
        ob = np.stack([(folded[a] ^ folded[b])[frm:to] for a, b in sub])
        hb = np.stack([(folded[a] & folded[b])[frm:to] for a, b in sub])
        wpost, _ = O.decode_batch(pm, ob, hb, frm, to)
        np.testing.assert_array_equal(post[gi][:, :, :cnt], wpost[frm:to])
        assert not post[gi][:, :, cnt:].any()
        full = np.zeros((pm.S, pm.K, cnt), np.float32)
        full[frm:to] = wpost[frm:to]
        for v in range(cnt):
            want.append(O.ibd_scan_pair(pm, full, v, sfrm, sto, pair_ordinal=first + v))
    want = np.concatenate(want)
    assert want.size > 5
    _assert_records_equal(got, want)
