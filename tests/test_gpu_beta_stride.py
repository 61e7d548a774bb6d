"""Beta stride of the IBD decode (fsmc_ctx_set_beta_stride): with stride 2 only every second beta row of a chunk
goes through HBM and the alpha sweep recomputes the others from their successor.  Every floating-point operation of
every row is the same as with stride 1, so the records must be identical to the oracle's bit for bit -- in the
single-chunk layout, in the checkpoint/recompute layout, for windows of even and odd length, for windows shorter
than a pair of sites, and with the per-state sums of open segments (mean / MAP ages) kept in the workspace."""
import numpy as np
import pytest

from fastsmc_amd import capi
from oracle import oracle as O

pytestmark = pytest.mark.gpu

FIELDS = (("pair", "pair"), ("start", "start"), ("end", "end"), ("prob", "prob"), ("post_mean", "postMean"),
          ("map", "map"))


def _pairs_array(pairs):
    return np.array(pairs, dtype=np.uint32).view(capi.PAIR_DTYPE).reshape(-1)


def _assert_records_equal(got, want):
    assert got.size == want.size
    for f_got, f_want in FIELDS:
        np.testing.assert_array_equal(got[f_got], want[f_want], err_msg=f_got)


def _ctx(sp, stride, limit=0, chunk=0):
    ctx = capi.Context(0)
    ctx.set_beta_stride(stride)
    if limit:
        ctx.set_workspace_limit(limit)
    if chunk:
        ctx.set_chunk_sites(chunk)
    model = ctx.create_model(sp["model"])
    ctx.upload_haps(sp["bits"], sp["model"].S)
    return ctx, model


@pytest.mark.parametrize("stride", [1, 2])
@pytest.mark.parametrize("flags", [capi.FSMC_WANT_MEAN | capi.FSMC_WANT_MAP, 0])
@pytest.mark.parametrize("limit", [0, 6 << 20])
def test_records_identical_to_oracle(small_problem, stride, flags, limit):
    pm = small_problem["model"]
    pairs = O.enumerate_all_pairs(32)[:200]  # three full groups and a ragged one
    want = O.decode_pairs_ibd(pm, small_problem["folded"], pairs, batch_size=64,
                              want_mean=bool(flags & capi.FSMC_WANT_MEAN), want_map=bool(flags & capi.FSMC_WANT_MAP))
    ctx, model = _ctx(small_problem, stride, limit)
    got = ctx.decode_ibd(model, _pairs_array(pairs), capi.whole_sequence_groups(len(pairs), pm.S), flags)
    assert ctx.last_beta_stride() == stride
    if limit:
        assert ctx.info()["max_chunks"] > 1
    ctx.close()
    assert want.size > 20
    _assert_records_equal(got, want)


def _window_oracle(sp, pairs, frm, to, sfrm, sto):
    pm = sp["model"]
    folded = sp["folded"]
    ob = np.stack([(folded[a] ^ folded[b])[frm:to] for a, b in pairs])
    hb = np.stack([(folded[a] & folded[b])[frm:to] for a, b in pairs])
    post, _ = O.decode_batch(pm, ob, hb, frm, to)
    full = np.zeros((pm.S, pm.K, len(pairs)), np.float32)
    full[frm:to] = post[frm:to]
    return full


@pytest.mark.parametrize("chunk", [0, 16, 48])
def test_odd_even_and_tiny_windows(small_problem, chunk):
    """Windows whose length, and whose last chunk's length, are odd, even, 1, 2 and 3 sites; scan windows that start
    and end inside a pair of sites.  Stride 2 against stride 1 and against the oracle."""
    pm = small_problem["model"]
    allp = O.enumerate_all_pairs(32)
    wins = [  # first pair, pairs, from, to, scan_from, scan_to
        (0, 64, 0, 640, 0, 640), (64, 40, 3, 636, 3, 636), (104, 64, 10, 331, 37, 300), (168, 9, 100, 101, 100, 101),
        (177, 33, 200, 202, 200, 202), (210, 64, 300, 303, 301, 303), (274, 64, 5, 422, 6, 421),
        (338, 20, 0, 49, 0, 49), (358, 64, 590, 640, 601, 640),
    ]
    n = wins[-1][0] + wins[-1][1]
    pairs = allp[1000:1000 + n]
    groups = np.zeros(len(wins), capi.GROUP_DTYPE)
    for g, w in zip(groups, wins):
        g["first_pair"], g["n_pairs"], g["from"], g["to"], g["scan_from"], g["scan_to"] = w
    flags = capi.FSMC_WANT_MEAN | capi.FSMC_WANT_MAP
    res = {}
    for stride in (1, 2):
        # 16 MB: the long windows are chunked, those no longer than a chunk take the single-chunk layout in the
        # same launch
        ctx, model = _ctx(small_problem, stride, limit=(16 << 20) if chunk else 0, chunk=chunk)
        res[stride] = ctx.decode_ibd(model, _pairs_array(pairs), groups, flags)
        assert (ctx.info()["max_chunks"] > 1) == bool(chunk)
        ctx.close()
    for f, _ in FIELDS:
        np.testing.assert_array_equal(res[2][f], res[1][f], err_msg=f)
    want = []
    for first, cnt, frm, to, sfrm, sto in wins:
        full = _window_oracle(small_problem, pairs[first:first + cnt], frm, to, sfrm, sto)
        for v in range(cnt):
            want.append(O.ibd_scan_pair(pm, full, v, sfrm, sto, pair_ordinal=first + v))
    want = np.concatenate(want)
    assert want.size > 20
    _assert_records_equal(res[2], want)


def test_stride_is_validated(small_problem):
    ctx = capi.Context(0)
    with pytest.raises(capi.FsmcError):
        ctx.set_beta_stride(3)
    ctx.close()


@pytest.mark.parametrize("K_time", [(69, 50), (50, 50), (33, 200), (100, 200)])
def test_sums_over_pairs_with_both_strides_and_resident_chunks(small_problem, K_time):
    """The sums over pairs have beta stride 2 and resident chunks too (round 5: at size they moved 8K bytes a pair-site
    at the rate the CUs' path to memory delivers).  The one-wave kernel (two-wave windows switched off): strides 1 and 2,
    whole windows and the checkpoint / rebuild layout with 0, 1 and every chunk resident, the 00 / 01 / 11 split, a
    ragged batch -- every variant the oracle's sums, bit for bit."""
    from fastsmc_amd import synth

    K, time = K_time
    if K == 69:
        pm, bits, folded = small_problem["model"], small_problem["bits"], small_problem["folded"]
    else:
        tables = synth.make_model_tables(K)
        haps = synth.make_haps(64, 333, seed=5, cm_per_mb=25.0, switch_per_cm=0.6)
        bits, derived, flipped = synth.fold_and_pack(haps.alleles)
        folded = np.where(flipped[None, :], 1 - haps.alleles, haps.alleles).astype(np.uint8)
        pm = O.prepare_model(tables, (haps.cm / 100.0).astype(np.float32), haps.bp, derived, 64, time=time)
    pairs = O.enumerate_all_pairs(32)[:151]
    want = [np.zeros((pm.S, pm.K), np.float32) for _ in range(4)]
    for b0 in range(0, len(pairs), 64):
        sub = pairs[b0:b0 + 64]
        ob = np.stack([folded[a] ^ folded[b] for a, b in sub])
        hb = np.stack([folded[a] & folded[b] for a, b in sub])
        post, _ = O.decode_batch(pm, ob, hb, 0, pm.S)
        O.augment_sum_over_pairs(pm, post, len(sub), ob, hb, *want)
    seen = set()
    for stride in (1, 2):
        for limit, chunk, resident in ((0, 0, -1), (1 << 30, 48, 0), (1 << 30, 48, 1), (1 << 30, 48, -1), (1 << 30, 47, -1)):
            ctx = capi.Context(0)
            ctx.set_two_wave_windows(1)
            ctx.set_beta_stride(stride)
            if limit:
                ctx.set_workspace_limit(limit)
            if chunk:
                ctx.set_chunk_sites(chunk)
            ctx.set_resident_chunks(resident)
            model = ctx.create_model(pm)
            ctx.upload_haps(bits, pm.S)
            ctx.upload_worklist(_pairs_array(pairs), capi.whole_sequence_groups(len(pairs), pm.S))
            s, mm = ctx.decode_sums(model, major_minor=True)
            # (the exact 50-state member has no stride-2 sums: its instantiation fails the in-flight check, fsmc_instances.h)
            assert ctx.last_beta_stride() == (stride if K != 50 else 1) and ctx.last_waves_per_window() == 1
            seen.add((stride, ctx.info()["max_chunks"] > 1, ctx.last_resident_chunks()))
            s_only, _ = ctx.decode_sums(model)
            ctx.close()
            np.testing.assert_array_equal(s, want[0], err_msg=f"stride {stride} chunk {chunk} resident {resident}")
            for got, w in zip(mm, want[1:]):
                np.testing.assert_array_equal(got, w)
            np.testing.assert_array_equal(s_only, want[0])
    assert any(c and r == 0 for _, c, r in seen) and any(c and r > 1 for _, c, r in seen) and any(not c for _, c, _ in seen)
