"""Two waves per decode window (csrc/fsmc_kernels_bidir.h): the consumers without state across sites -- posterior dump,
per-pair mean / MAP rows, sums over pairs -- of a small launch run alpha up from the window's first site in one wave
while beta comes down from its last in another.  Every posterior is the one-wave kernel's, bit for bit, and the oracle's:
windows of one, two, three sites, odd and even lengths, sub-windows, ragged groups, batches of more than 64 pairs (the
sums' rounds), the 00 / 01 / 11 split, padded and exact members with one and two waves per SIMD."""
import numpy as np
import pytest

from fastsmc_amd import capi, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _problem(K, S=150, seed=3):
    tables = synth.make_model_tables(K)
    haps = synth.make_haps(64, S, seed=seed, cm_per_mb=25.0, switch_per_cm=0.6)
    bits, derived, flipped = synth.fold_and_pack(haps.alleles)
    folded = np.where(flipped[None, :], 1 - haps.alleles, haps.alleles).astype(np.uint8)
    pm = O.prepare_model(tables, (haps.cm / 100.0).astype(np.float32), haps.bp, derived, 64, time=200)
    return pm, bits, folded


def _pairs_array(pairs):
    return np.array(pairs, dtype=np.uint32).view(capi.PAIR_DTYPE).reshape(-1)


@pytest.mark.parametrize("K", [69, 50, 16, 33, 64, 80, 81, 99, 100, 112, 120, 128])
def test_whole_windows_every_consumer(K):
    two = 2
    pm, bits, folded = _problem(K)
    pairs = O.enumerate_all_pairs(32)[:151]  # 64 + 64 + 23
    pr = _pairs_array(pairs)
    groups = capi.whole_sequence_groups(len(pairs), pm.S)
    got = {}
    for mode in (0, 1):  # automatic (this launch qualifies), never
        ctx = capi.Context(0)
        ctx.set_two_wave_windows(mode)
        model = ctx.create_model(pm)
        ctx.upload_haps(bits, pm.S)
        ctx.upload_worklist(pr, groups)
        post = ctx.decode_posteriors(model)
        w_dump = ctx.last_waves_per_window()
        mean, mp = ctx.decode_per_pair(model, pm.exp_times)
        w_pp = ctx.last_waves_per_window()
        s, mm = ctx.decode_sums(model, major_minor=True)
        w_sums = ctx.last_waves_per_window()
        s_only, _ = ctx.decode_sums(model)
        assert (w_dump, w_pp, w_sums) == ((two, two, two) if mode == 0 else (1, 1, 1))
        # a second run of every consumer returns the same bits (no race between the two waves of a window)
        for a, b in zip(post, ctx.decode_posteriors(model)):
            np.testing.assert_array_equal(a, b)
        mean2, mp2 = ctx.decode_per_pair(model, pm.exp_times)
        np.testing.assert_array_equal(mean, mean2)
        np.testing.assert_array_equal(mp, mp2)
        np.testing.assert_array_equal(s_only, ctx.decode_sums(model)[0])
        got[mode] = (post, mean, mp, s, mm, s_only)
        ctx.close()
    for a, b in zip(got[0][0], got[1][0]):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(got[0][1], got[1][1])
    np.testing.assert_array_equal(got[0][2], got[1][2])
    np.testing.assert_array_equal(got[0][3], got[1][3])
    for a, b in zip(got[0][4], got[1][4]):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(got[0][5], got[1][5])
    # ... and the oracle's
    want = [np.zeros((pm.S, pm.K), np.float32) for _ in range(4)]
    for gi in range(3):
        sub = pairs[64 * gi:64 * gi + 64]
        ob = np.stack([folded[a] ^ folded[b] for a, b in sub])
        hb = np.stack([folded[a] & folded[b] for a, b in sub])
        wpost, _ = O.decode_batch(pm, ob, hb, 0, pm.S)
        np.testing.assert_array_equal(got[0][0][gi][:, :, :len(sub)], wpost)
        wmean, wmap, _ = O.per_pair_output(pm, wpost, len(sub))
        np.testing.assert_array_equal(got[0][1][64 * gi:64 * gi + len(sub)], wmean)
        np.testing.assert_array_equal(got[0][2][64 * gi:64 * gi + len(sub)], wmap)
        O.augment_sum_over_pairs(pm, wpost, len(sub), ob, hb, *want)
    np.testing.assert_array_equal(got[0][3], want[0])
    for a, b in zip(got[0][4], want[1:]):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(got[0][5], want[0])


def test_short_and_odd_windows():
    """Windows of 1, 2, 3, 4, 5 sites and a few longer odd / even ones, inside the sequence: dump and per-pair rows."""
    pm, bits, folded = _problem(69, S=130)
    allp = O.enumerate_all_pairs(32)
    wins = [(0, 3, 7, 8), (3, 64, 20, 22), (67, 5, 40, 43), (72, 64, 0, 4), (136, 9, 125, 130), (145, 33, 1, 130),
            (178, 64, 63, 128), (242, 2, 64, 65)]
    n = wins[-1][0] + wins[-1][1]
    pairs = allp[300:300 + n]
    groups = np.zeros(len(wins), capi.GROUP_DTYPE)
    for g, (first, cnt, frm, to) in zip(groups, wins):
        g["first_pair"], g["n_pairs"], g["from"], g["to"], g["scan_from"], g["scan_to"] = first, cnt, frm, to, frm, to
    res = {}
    for mode in (0, 1):
        ctx = capi.Context(0)
        ctx.set_two_wave_windows(mode)
        model = ctx.create_model(pm)
        ctx.upload_haps(bits, pm.S)
        ctx.upload_worklist(_pairs_array(pairs), groups)
        post = ctx.decode_posteriors(model)
        assert ctx.last_waves_per_window() == (2 if mode == 0 else 1)
        mean, mp = ctx.decode_per_pair(model, pm.exp_times)
        res[mode] = (post, mean, mp)
        ctx.close()
    for gi, (first, cnt, frm, to) in enumerate(wins):
        sub = pairs[first:first + cnt]
        ob = np.stack([(folded[a] ^ folded[b])[frm:to] for a, b in sub])
        hb = np.stack([(folded[a] & folded[b])[frm:to] for a, b in sub])
        wpost, _ = O.decode_batch(pm, ob, hb, frm, to)
        for mode in (0, 1):
            np.testing.assert_array_equal(res[mode][0][gi][:, :, :cnt], wpost[frm:to], err_msg=f"group {gi} mode {mode}")
            assert not res[mode][0][gi][:, :, cnt:].any()
    np.testing.assert_array_equal(res[0][1], res[1][1])
    np.testing.assert_array_equal(res[0][2], res[1][2])
    # the rows of sites outside a pair's window stay zero, inside they are the oracle's
    first, cnt, frm, to = wins[5]
    sub = pairs[first:first + cnt]
    ob = np.stack([(folded[a] ^ folded[b])[frm:to] for a, b in sub])
    hb = np.stack([(folded[a] & folded[b])[frm:to] for a, b in sub])
    wpost, _ = O.decode_batch(pm, ob, hb, frm, to)
    full = np.zeros((pm.S, pm.K, cnt), np.float32)
    full[frm:to] = wpost[frm:to]
    wmean, wmap, _ = O.per_pair_output(pm, full, cnt)
    np.testing.assert_array_equal(res[0][1][first:first + cnt, frm:to], wmean[:, frm:to])
    np.testing.assert_array_equal(res[0][2][first:first + cnt, frm:to], wmap[:, frm:to])


def test_sums_of_batches_of_more_than_64_pairs():
    """A reference batch of 160 pairs is three groups that share one running sum: the workgroup decodes them in turn and
    each of its two waves continues the sums of ITS half of the sites (the split site is the same in every round)."""
    pm, bits, folded = _problem(69, S=97)
    pairs = O.enumerate_all_pairs(32)[:400]  # batches of 160, 160, 80
    groups = []
    bfg = [0]
    for b0 in range(0, 400, 160):
        nb = min(160, 400 - b0)
        for g0 in range(0, nb, 64):
            groups.append((b0 + g0, min(64, nb - g0)))
        bfg.append(len(groups))
    gr = np.zeros(len(groups), capi.GROUP_DTYPE)
    for g, (first, cnt) in zip(gr, groups):
        g["first_pair"], g["n_pairs"], g["from"], g["to"], g["scan_from"], g["scan_to"] = first, cnt, 0, pm.S, 0, pm.S
    out = {}
    for mode in (0, 1):
        ctx = capi.Context(0)
        ctx.set_two_wave_windows(mode)
        model = ctx.create_model(pm)
        ctx.upload_haps(bits, pm.S)
        ctx.upload_worklist(_pairs_array(pairs), gr)
        s, mm = ctx.decode_sums(model, major_minor=True, batch_first_group=np.array(bfg))
        assert ctx.last_waves_per_window() == (2 if mode == 0 else 1)
        out[mode] = (s, mm)
        ctx.close()
    want = [np.zeros((pm.S, pm.K), np.float32) for _ in range(4)]
    for b0 in range(0, 400, 160):
        sub = pairs[b0:b0 + 160]
        ob = np.stack([folded[a] ^ folded[b] for a, b in sub])
        hb = np.stack([folded[a] & folded[b] for a, b in sub])
        wpost, _ = O.decode_batch(pm, ob, hb, 0, pm.S)  # (a batch of any size: lanes are independent)
        O.augment_sum_over_pairs(pm, wpost, len(sub), ob, hb, *want)
    for mode in (0, 1):
        np.testing.assert_array_equal(out[mode][0], want[0])
        for a, b in zip(out[mode][1], want[1:]):
            np.testing.assert_array_equal(a, b)


def test_large_launches_and_other_models_keep_one_wave():
    """More groups than half the chip's waves, sequence mode, wide models: the one-wave kernels."""
    pm, bits, _ = _problem(69, S=64)
    ctx = capi.Context(0)
    model = ctx.create_model(pm)
    ctx.upload_haps(bits, pm.S)
    n_slots = 8 * ctx.info()["n_cu"]  # the 69-state member runs two waves per SIMD: eight a CU
    pairs = np.tile(np.array(O.enumerate_all_pairs(32)[:64], np.uint32), (n_slots // 2 + 1, 1))
    ctx.upload_worklist(pairs.view(capi.PAIR_DTYPE).reshape(-1), capi.whole_sequence_groups(pairs.shape[0], pm.S))
    ctx.decode_sums(model)
    assert ctx.last_waves_per_window() == 1
    ctx.upload_worklist(pairs[:64 * (n_slots // 2)].view(capi.PAIR_DTYPE).reshape(-1),
                        capi.whole_sequence_groups(64 * (n_slots // 2), pm.S))
    ctx.decode_sums(model)
    assert ctx.last_waves_per_window() == 2
    ctx.close()
    pm256, bits256, _ = _problem(256, S=64)
    ctx = capi.Context(0)
    model = ctx.create_model(pm256)
    ctx.upload_haps(bits256, pm256.S)
    ctx.upload_worklist(_pairs_array(O.enumerate_all_pairs(32)[:64]), capi.whole_sequence_groups(64, pm256.S))
    ctx.decode_sums(model)
    assert ctx.last_waves_per_window() == 1
    ctx.close()
