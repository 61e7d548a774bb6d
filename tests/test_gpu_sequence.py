"""Sequence mode (DecodingParams::decodingSequence, scope row f4) on the GPU against the oracle -- bit-exact in
every output mode: two transition steps per site (homozygous stretch, then the site; HMM.cpp:760-770, 915-925) and
the posterior built from the vectors the reference's buffers hold after its in-place copies (oracle/hmm_oracle.h)."""
import numpy as np
import pytest

from conftest import expected_member, wave_group_member
from fastsmc_amd import api, capi, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _pairs_array(pairs):
    return np.array(pairs, dtype=np.uint32).view(capi.PAIR_DTYPE).reshape(-1)


def _oracle_posterior(sp, pm, pairs, frm, to):
    folded = sp["folded"]
    ob = np.stack([(folded[a] ^ folded[b])[frm:to] for a, b in pairs])
    hb = np.stack([(folded[a] & folded[b])[frm:to] for a, b in pairs])
    post, _ = O.decode_batch(pm, ob, hb, frm, to)
    return post


def _assert_records_equal(got, want):
    assert got.size == want.size
    for f_got, f_want in (("pair", "pair"), ("start", "start"), ("end", "end"), ("prob", "prob"),
                          ("post_mean", "postMean"), ("map", "map")):
        np.testing.assert_array_equal(got[f_got], want[f_want], err_msg=f_got)


@pytest.fixture(scope="module")
def gpu(seq_problem):
    ctx = capi.Context(0)
    model = ctx.create_model(seq_problem["model"])
    ctx.upload_haps(seq_problem["bits"], seq_problem["model"].S)
    yield ctx, model
    ctx.close()


def test_posterior_whole_sequence_and_windows(gpu, seq_problem):
    ctx, model = gpu
    pm = seq_problem["model"]
    pairs = O.enumerate_all_pairs(32)[:100]
    ctx.upload_worklist(_pairs_array(pairs), capi.whole_sequence_groups(len(pairs), pm.S))
    got = ctx.decode_posteriors(model)
    for gi, (lo, n) in enumerate(((0, 64), (64, 36))):
        want = _oracle_posterior(seq_problem, pm, pairs[lo:lo + n], 0, pm.S)
        np.testing.assert_array_equal(got[gi][:, :, :n], want)
    # windows, including one- and two-site ones (no half-step at all / exactly one per direction)
    groups = np.zeros(4, capi.GROUP_DTYPE)
    wins = [(0, 17, 100, 311), (17, 3, 5, 6), (20, 20, 398, 400), (40, 9, 0, 2)]
    for g, (first, n, frm, to) in zip(groups, wins):
        g["first_pair"], g["n_pairs"], g["from"], g["to"], g["scan_from"], g["scan_to"] = first, n, frm, to, frm, to
    ctx.upload_worklist(_pairs_array(pairs[:49]), groups)
    got = ctx.decode_posteriors(model)
    for gi, (first, n, frm, to) in enumerate(wins):
        want = _oracle_posterior(seq_problem, pm, pairs[first:first + n], frm, to)[frm:to]
        np.testing.assert_array_equal(got[gi][:, :, :n], want)


@pytest.mark.parametrize("flags", [capi.FSMC_WANT_MEAN | capi.FSMC_WANT_MAP, 0])
def test_ibd_records(gpu, seq_problem, flags):
    ctx, model = gpu
    pm = seq_problem["model"]
    pairs = O.enumerate_all_pairs(32)[:200]
    want = O.decode_pairs_ibd(pm, seq_problem["folded"], pairs, batch_size=64,
                              want_mean=bool(flags & capi.FSMC_WANT_MEAN), want_map=bool(flags & capi.FSMC_WANT_MAP))
    got = ctx.decode_ibd(model, _pairs_array(pairs), capi.whole_sequence_groups(len(pairs), pm.S), flags)
    assert want.size > 20
    _assert_records_equal(got, want)


def test_scan_window_and_chunked_stream(seq_problem):
    """Checkpointed beta stream (tiny workspace => 16/32-site chunks) with a scan window inside the decode window."""
    pm = seq_problem["model"]
    pairs = O.enumerate_all_pairs(32)[300:300 + 100]
    frm, to, sfrm, sto = 13, 390, 40, 377
    groups = np.zeros(2, capi.GROUP_DTYPE)
    groups[0] = (0, 64, frm, to, sfrm, sto)
    groups[1] = (64, 36, 0, pm.S, 0, pm.S)
    want = []
    for first, n, f, t, sf, st in [tuple(int(x) for x in g) for g in groups]:
        post = _oracle_posterior(seq_problem, pm, pairs[first:first + n], f, t)
        want += [O.ibd_scan_pair(pm, post, v, sf, st, pair_ordinal=first + v) for v in range(n)]
    want = np.concatenate(want)
    assert want.size > 10
    plans = []
    for limit, chunk in ((0, 0), (6 << 20, 0), (12 << 20, 16)):
        ctx = capi.Context(0)
        if limit:
            ctx.set_workspace_limit(limit)
        if chunk:
            ctx.set_chunk_sites(chunk)
        model = ctx.create_model(pm)
        ctx.upload_haps(seq_problem["bits"], pm.S)
        got = ctx.decode_ibd(model, _pairs_array(pairs), groups)
        plans.append(ctx.info()["chunk_sites"])
        ctx.upload_worklist(_pairs_array(pairs), groups)
        post = ctx.decode_posteriors(model)
        ctx.close()
        _assert_records_equal(got, want)
        np.testing.assert_array_equal(post[1][:, :, :36], _oracle_posterior(seq_problem, pm, pairs[64:], 0, pm.S))
    # whole window in one chunk; the largest chunk the 6-MB workspace allows (the window is chunked); the forced 16
    assert plans[0] >= pm.S - 13 and plans[1] < 200 and plans[2] == 16


def test_per_pair_and_sums(gpu, seq_problem):
    ctx, model = gpu
    pm = seq_problem["model"]
    folded = seq_problem["folded"]
    pairs = O.enumerate_all_pairs(32)[100:100 + 150]
    ctx.upload_worklist(_pairs_array(pairs), capi.whole_sequence_groups(len(pairs), pm.S))
    mean, mp = ctx.decode_per_pair(model, pm.exp_times)
    s, (s00, s01, s11) = ctx.decode_sums(model, major_minor=True)
    want = [np.zeros((pm.S, pm.K), np.float32) for _ in range(4)]
    for b0 in range(0, len(pairs), 64):
        chunk = pairs[b0:b0 + 64]
        ob = np.stack([folded[a] ^ folded[b] for a, b in chunk])
        hb = np.stack([folded[a] & folded[b] for a, b in chunk])
        post, _ = O.decode_batch(pm, ob, hb, 0, pm.S)
        wmean, wmap, _ = O.per_pair_output(pm, post, len(chunk))
        np.testing.assert_array_equal(mean[b0:b0 + len(chunk)], wmean)
        np.testing.assert_array_equal(mp[b0:b0 + len(chunk)], wmap)
        O.augment_sum_over_pairs(pm, post, len(chunk), ob, hb, want[0], want[1], want[2], want[3])
    for got, w in zip((s, s00, s01, s11), want):
        np.testing.assert_array_equal(got, w)


@pytest.mark.parametrize("K", [12, 100, 150, 200, 256, 260, 340, 400, 460, 520, 650, 800, 1030])
def test_generic_k_sequence(K):
    tables = synth.make_model_tables(K)
    haps = synth.make_haps(64, 150, seed=5, cm_per_mb=1.2, bp_per_site=2500, switch_per_cm=2.0)
    bits, derived, flipped = synth.fold_and_pack(haps.alleles)
    folded = np.where(flipped[None, :], 1 - haps.alleles, haps.alleles).astype(np.uint8)
    gen = (haps.cm / 100.0).astype(np.float32)
    pm = O.prepare_model(tables, gen, haps.bp, derived, 64, time=200, decoding_sequence=True)
    sp = dict(folded=folded)
    pairs = O.enumerate_all_pairs(32)[:80]
    ctx = capi.Context(0)
    model = ctx.create_model(pm)
    ctx.upload_haps(bits, pm.S)
    groups = capi.whole_sequence_groups(len(pairs), pm.S)
    _assert_records_equal(ctx.decode_ibd(model, _pairs_array(pairs), groups),
                          O.decode_pairs_ibd(pm, folded, pairs, batch_size=64))
    if K > 128:  # the wave-group kernel up to 1024 states, the any-K kernel beyond, also in sequence mode
        assert ctx.last_kernel() == expected_member(K)
    ctx.upload_worklist(_pairs_array(pairs), groups)
    post = ctx.decode_posteriors(model)
    wpost = _oracle_posterior(sp, pm, pairs[:64], 0, pm.S)
    np.testing.assert_array_equal(post[0], wpost)
    if K > 128:
        # the other consumers of the wave-group kernel in sequence mode, and the chunked beta stream
        ctx.upload_worklist(_pairs_array(pairs[:64]), capi.whole_sequence_groups(64, pm.S))
        mean, mp = ctx.decode_per_pair(model, pm.exp_times)
        wmean, wmap, _ = O.per_pair_output(pm, wpost, 64)
        np.testing.assert_array_equal(mean, wmean)
        np.testing.assert_array_equal(mp, wmap)
        ob = np.stack([folded[a] ^ folded[b] for a, b in pairs[:64]])
        hb = np.stack([folded[a] & folded[b] for a, b in pairs[:64]])
        s_, _ = ctx.decode_sums(model)
        wsum = np.zeros((pm.S, pm.K), np.float32)
        O.augment_sum_over_pairs(pm, wpost, 64, ob, hb, wsum)
        np.testing.assert_array_equal(s_, wsum)
        ctx.set_chunk_sites(32)
        nw, kh = wave_group_member(K)
        row_bytes = (nw * kh if K <= 1024 else (K + 15) // 16 * 16) * 256
        ctx.set_workspace_limit(60 * row_bytes * 2)  # ~60 rows for each of the two groups: 150 sites do not fit
        _assert_records_equal(ctx.decode_ibd(model, _pairs_array(pairs), groups),
                              O.decode_pairs_ibd(pm, folded, pairs, batch_size=64))
        assert ctx.info()["max_chunks"] > 1
    ctx.close()


def test_asmc_api_in_sequence_mode(seq_problem, tmp_path):
    """ASMC(params with mode "sequence").decodePairs through files (ASMC.cpp:80-128): the product's host code picks
    the sequence-mode emissions and rows, the kernel the two-step recursion."""
    sp = seq_problem
    root = str(tmp_path / "seq")
    synth.write_haps_files(root, sp["haps"], fastsmc_map=False)
    synth.write_decoding_quantities(root + ".decodingQuantities.gz", sp["tables"])
    p = api.DecodingParams(root, root + ".decodingQuantities.gz", decodingModeString="sequence",
                           doPerPairPosteriorMean=True)
    p.doPerPairMAP = True
    p.useKnownSeed = True
    asmc = api.ASMC(p)
    a = [1, 2, 3, 10, 40, 63, 7]
    b = [2, 3, 4, 11, 41, 0, 9]
    asmc.decodePairs(a, b, True, True, True, True)
    res = asmc.get_copy_of_results()
    data = api.Data(p)
    gen = np.array(data.geneticPositions, np.float32)
    rate = np.array(data.recRateAtMarker, np.float32)
    _, derived, _ = synth.fold_and_pack(sp["haps"].alleles)
    pm = O.prepare_model(sp["tables"], gen, sp["haps"].bp, derived, 64, time=p.time, decoding_sequence=True,
                         rec_rate=rate, no_conditional_age_estimates=p.noConditionalAgeEstimates)
    folded = sp["folded"]
    ob = np.stack([folded[x] ^ folded[y] for x, y in zip(a, b)] + [folded[a[-1]] ^ folded[b[-1]]])
    hb = np.stack([folded[x] & folded[y] for x, y in zip(a, b)] + [folded[a[-1]] & folded[b[-1]]])
    post, _ = O.decode_batch(pm, ob, hb, 0, pm.S)
    sum_post = np.zeros((pm.K, pm.S), np.float32)
    wmean, wmap, wpost = O.per_pair_output(pm, post, len(a), want_post=True, sum_of_post=sum_post)
    np.testing.assert_array_equal(res.per_pair_posterior_means, wmean)
    np.testing.assert_array_equal(res.per_pair_MAPs, wmap)
    for i in range(len(a)):
        np.testing.assert_array_equal(res.per_pair_posteriors[i], wpost[i])
    np.testing.assert_array_equal(res.sum_of_posteriors, sum_post)


def test_sequence_mode_with_a_wide_model(seq_problem):
    """A 100-state model in sequence mode: the 112-state member of the lane-per-pair family (12 ghost states)."""
    sp = seq_problem
    tables = synth.make_model_tables(100)
    haps = sp["haps"]
    _, derived, _ = synth.fold_and_pack(haps.alleles)
    pm = O.prepare_model(tables, sp["gen"], haps.bp, derived, 64, time=200, decoding_sequence=True)
    pairs = O.enumerate_all_pairs(32)[:90]
    want = O.decode_pairs_ibd(pm, sp["folded"], pairs, batch_size=64)
    ctx = capi.Context(0)
    model = ctx.create_model(pm)
    ctx.upload_haps(sp["bits"], pm.S)
    got = ctx.decode_ibd(model, _pairs_array(pairs), capi.whole_sequence_groups(len(pairs), pm.S))
    ctx.upload_worklist(_pairs_array(pairs[:64]), capi.whole_sequence_groups(64, pm.S))
    post = ctx.decode_posteriors(model)[0]
    ctx.close()
    assert want.size > 5
    _assert_records_equal(got, want)
    np.testing.assert_array_equal(post, _oracle_posterior(sp, pm, pairs[:64], 0, pm.S))
