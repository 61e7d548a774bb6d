"""Parity of the HIP decode path with the CPU oracle, through the C ABI (include/fastsmc_hip.h).

Bar: bit-exact.  Posteriors, IBD coordinates and every float field of an IBD record must be identical
to the oracle's (the kernel evaluates every sum in the reference's order, without FMA)."""
import numpy as np
import pytest

from fastsmc_amd import capi
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(small_problem):
    ctx = capi.Context(0)
    model = ctx.create_model(small_problem["model"])
    ctx.upload_haps(small_problem["bits"], small_problem["model"].S)
    yield ctx, model
    ctx.close()


def _pairs_array(pairs):
    return np.array(pairs, dtype=np.uint32).view(capi.PAIR_DTYPE).reshape(-1)


def _oracle_posterior(sp, pairs, frm, to):
    folded = sp["folded"]
    ob = np.stack([(folded[a] ^ folded[b])[frm:to] for a, b in pairs])
    hb = np.stack([(folded[a] & folded[b])[frm:to] for a, b in pairs])
    post, _ = O.decode_batch(sp["model"], ob, hb, frm, to)
    return post[frm:to]  # [to-frm][K][B]


def _assert_records_equal(got, want):
    assert got.size == want.size
    for f_got, f_want in (("pair", "pair"), ("start", "start"), ("end", "end"), ("prob", "prob"),
                          ("post_mean", "postMean"), ("map", "map")):
        np.testing.assert_array_equal(got[f_got], want[f_want], err_msg=f_got)


def test_posterior_bit_exact_whole_sequence(gpu, small_problem):
    ctx, model = gpu
    S = small_problem["model"].S
    pairs = O.enumerate_all_pairs(32)[:100]  # 64 + 36: a full and a ragged group
    ctx.upload_worklist(_pairs_array(pairs), capi.whole_sequence_groups(len(pairs), S))
    got = ctx.decode_posteriors(model)
    for gi, (lo, n) in enumerate(((0, 64), (64, 36))):
        want = _oracle_posterior(small_problem, pairs[lo:lo + n], 0, S)
        np.testing.assert_array_equal(got[gi][:, :, :n], want)
        assert not got[gi][:, :, n:].any()
        np.testing.assert_allclose(got[gi][:, :, :n].sum(axis=1), 1.0, rtol=1e-5)


def test_posterior_bit_exact_sub_windows(gpu, small_problem):
    """Hashing-mode shape: every group has its own decode window."""
    ctx, model = gpu
    pairs = O.enumerate_all_pairs(32)[200:200 + 40]
    groups = np.zeros(3, capi.GROUP_DTYPE)
    wins = [(0, 17, 100, 400), (17, 3, 0, 2), (20, 20, 600, 640)]
    for g, (first, n, frm, to) in zip(groups, wins):
        g["first_pair"], g["n_pairs"], g["from"], g["to"], g["scan_from"], g["scan_to"] = first, n, frm, to, frm, to
    ctx.upload_worklist(_pairs_array(pairs), groups)
    got = ctx.decode_posteriors(model)
    for gi, (first, n, frm, to) in enumerate(wins):
        want = _oracle_posterior(small_problem, pairs[first:first + n], frm, to)
        np.testing.assert_array_equal(got[gi][:, :, :n], want)


@pytest.mark.parametrize("flags", [capi.FSMC_WANT_MEAN | capi.FSMC_WANT_MAP, 0, capi.FSMC_WANT_MAP])
def test_ibd_records_identical(gpu, small_problem, flags):
    ctx, model = gpu
    pm = small_problem["model"]
    pairs = O.enumerate_all_pairs(32)[:200]
    want = O.decode_pairs_ibd(pm, small_problem["folded"], pairs, batch_size=64,
                              want_mean=bool(flags & capi.FSMC_WANT_MEAN), want_map=bool(flags & capi.FSMC_WANT_MAP))
    got = ctx.decode_ibd(model, _pairs_array(pairs), capi.whole_sequence_groups(len(pairs), pm.S), flags)
    assert want.size > 20  # the fixture must actually exercise the record path
    _assert_records_equal(got, want)


def test_batch_size_32_groups_match_reference_batches(gpu, small_problem):
    """FastSMC's default batch is 32 pairs (DecodingParams.cpp:61): groups of 32 give the same records."""
    ctx, model = gpu
    pm = small_problem["model"]
    pairs = O.enumerate_all_pairs(32)[300:300 + 70]
    want = O.decode_pairs_ibd(pm, small_problem["folded"], pairs, batch_size=32)
    got = ctx.decode_ibd(model, _pairs_array(pairs), capi.whole_sequence_groups(len(pairs), pm.S, batch=32))
    _assert_records_equal(got, want)


def test_chunked_beta_stream_is_identical(small_problem):
    """Force the checkpoint/recompute path (tiny workspace) and compare with the single-chunk result."""
    pm = small_problem["model"]
    pairs = O.enumerate_all_pairs(32)[:130]
    pr = _pairs_array(pairs)
    groups = capi.whole_sequence_groups(len(pairs), pm.S)
    results = []
    for limit in (0, 4 << 20):  # default (single chunk, 36 MB) vs 4 MB: forces 32-site chunks + checkpoints
        ctx = capi.Context(0)
        if limit:
            ctx.set_workspace_limit(limit)
        model = ctx.create_model(pm)
        ctx.upload_haps(small_problem["bits"], pm.S)
        rec = ctx.decode_ibd(model, pr, groups)
        ctx.upload_worklist(pr, groups)
        post = ctx.decode_posteriors(model)
        results.append((rec, post))
        ctx.close()
    _assert_records_equal(results[1][0], _as_oracle(results[0][0]))
    for a, b in zip(results[0][1], results[1][1]):
        np.testing.assert_array_equal(a, b)


def _as_oracle(rec):
    out = np.zeros(rec.size, O.IBD_DTYPE)
    for f_got, f_want in (("pair", "pair"), ("start", "start"), ("end", "end"), ("prob", "prob"),
                          ("post_mean", "postMean"), ("map", "map")):
        out[f_want] = rec[f_got]
    return out


def test_scan_window_inside_decode_window(gpu, small_problem):
    """Hashing mode scans [scan_from, scan_to) inside the padded decode window (HMM.cpp:1199-1206)."""
    ctx, model = gpu
    pm = small_problem["model"]
    pairs = O.enumerate_all_pairs(32)[500:500 + 64]
    frm, to, sfrm, sto = 50, 600, 120, 540
    groups = np.zeros(1, capi.GROUP_DTYPE)
    groups[0] = (0, 64, frm, to, sfrm, sto)
    got = ctx.decode_ibd(model, _pairs_array(pairs), groups)
    post = _oracle_posterior(small_problem, pairs, frm, to)
    full = np.zeros((pm.S, pm.K, 64), np.float32)
    full[frm:to] = post
    want = np.concatenate([O.ibd_scan_pair(pm, full, v, sfrm, sto, pair_ordinal=v) for v in range(64)])
    _assert_records_equal(got, want)


def test_bad_worklists_are_rejected_on_the_host(gpu, small_problem):
    ctx, model = gpu
    S = small_problem["model"].S
    pr = _pairs_array([(0, 1), (2, 3)])
    g = capi.whole_sequence_groups(2, S)
    bad = g.copy(); bad["n_pairs"] = 65
    with pytest.raises(capi.FsmcError):
        ctx.upload_worklist(pr, bad)
    bad = g.copy(); bad["to"] = 0
    with pytest.raises(capi.FsmcError):
        ctx.upload_worklist(pr, bad)
    bad = g.copy(); bad["to"] = S + 1; bad["scan_to"] = S + 1
    ctx.upload_worklist(pr, bad)
    with pytest.raises(capi.FsmcError):
        ctx.decode_ibd_launch(model)
    ctx.upload_worklist(_pairs_array([(0, 1), (2, 9999)]), g)
    with pytest.raises(capi.FsmcError):
        ctx.decode_ibd_launch(model)
