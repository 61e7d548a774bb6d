"""Models other than K = 69 against the oracle, all output modes.  K <= 80 runs the lane-per-pair kernel compiled for
the next family member (16, 32, 48, 64, 80 states with two waves per SIMD, 96, 112, 128 with one; the padding states are
ghosts -- K = 2, 5, 16 exactly, 17, 33, 50, 64 exactly, 65, 70, 80 exactly, 81, 100, 128 exactly); 128 < K <= 256 the
wide-model kernel, ghost-padded (K = 130, 192 exactly, 200, 256 exactly): four waves per group with lane = pair, every
consumer.  Every launch is checked for the family member it ran (fsmc_ctx_last_kernel)."""
import numpy as np
import pytest

from conftest import expected_member
from fastsmc_amd import capi, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _problem(K, n_hap=64, S=200, seed=11):
    tables = synth.make_model_tables(K)
    haps = synth.make_haps(n_hap, S, seed=seed, cm_per_mb=25.0, switch_per_cm=0.6)
    bits, derived, flipped = synth.fold_and_pack(haps.alleles)
    folded = np.where(flipped[None, :], 1 - haps.alleles, haps.alleles).astype(np.uint8)
    gen = (haps.cm / 100.0).astype(np.float32)
    pm = O.prepare_model(tables, gen, haps.bp, derived, n_hap, time=200)
    return pm, bits, folded


def _member(K, consumer="ibd"):
    return expected_member(K)  # (conftest.py)


def _stride(K):
    return 2 if K <= 128 else 1  # every member of the lane-per-pair family is built with beta stride 2


@pytest.mark.parametrize("K", [2, 5, 16, 17, 33, 49, 50, 51, 64, 65, 70, 80, 81, 99, 100, 101, 128, 130, 192, 200, 256, 257, 300, 320, 321,
                               384, 385, 402, 448, 449, 512, 513, 520, 560, 561, 600, 640, 641, 700, 768, 769, 1000, 1024,
                               1025])
def test_generic_kernel_matches_oracle(K):
    pm, bits, folded = _problem(K)
    pairs = O.enumerate_all_pairs(32)[:96]
    pr = np.array(pairs, dtype=np.uint32).view(capi.PAIR_DTYPE).reshape(-1)
    ctx = capi.Context(0)
    model = ctx.create_model(pm)
    ctx.upload_haps(bits, pm.S)
    groups = capi.whole_sequence_groups(len(pairs), pm.S)
    # IBD records (mean + MAP)
    want = O.decode_pairs_ibd(pm, folded, pairs, batch_size=64)
    got = ctx.decode_ibd(model, pr, groups)
    assert ctx.last_kernel() == _member(K)
    assert ctx.last_beta_stride() == _stride(K)
    assert got.size == want.size
    for f_got, f_want in (("pair", "pair"), ("start", "start"), ("end", "end"), ("prob", "prob"),
                          ("post_mean", "postMean"), ("map", "map")):
        np.testing.assert_array_equal(got[f_got], want[f_want], err_msg=f_got)
    # (K <= 128, array mode: once with two waves per window -- this launch qualifies -- and once on the one-wave kernel)
    for two_wave in ((0, 1) if K <= 128 else (0,)):
        ctx.set_two_wave_windows(two_wave)
        # posterior, per-pair mean/MAP and sums for the first group
        ctx.upload_worklist(pr[:64], capi.whole_sequence_groups(64, pm.S))
        post = ctx.decode_posteriors(model)[0]
        assert ctx.last_kernel() == _member(K)
        ob = np.stack([folded[a] ^ folded[b] for a, b in pairs[:64]])
        hb = np.stack([folded[a] & folded[b] for a, b in pairs[:64]])
        wpost, _ = O.decode_batch(pm, ob, hb, 0, pm.S)
        np.testing.assert_array_equal(post, wpost)
        mean, mp = ctx.decode_per_pair(model, pm.exp_times)
        assert ctx.last_kernel() == _member(K, "per_pair")
        wmean, wmap, _ = O.per_pair_output(pm, wpost, 64)
        np.testing.assert_array_equal(mean, wmean)
        np.testing.assert_array_equal(mp, wmap)
        s, _ = ctx.decode_sums(model)  # (K = 256 included: the transposition tile is sized for it)
        assert ctx.last_waves_per_window() == (2 if K <= 128 and two_wave == 0 else 1)
        assert ctx.last_kernel() == _member(K)
        wsum = np.zeros((pm.S, pm.K), np.float32)
        O.augment_sum_over_pairs(pm, wpost, 64, ob, hb, wsum)
        np.testing.assert_array_equal(s, wsum)
        if K in (5, 100, 200, 256, 300, 402, 520, 700, 1000, 1025):  # the 00 / 01 / 11 split in one member of each kernel
            s2, mm = ctx.decode_sums(model, major_minor=True)
            want = [np.zeros((pm.S, pm.K), np.float32) for _ in range(4)]
            O.augment_sum_over_pairs(pm, wpost, 64, ob, hb, want[0], want[1], want[2], want[3])
            for got, w in zip((s2, *mm), want):
                np.testing.assert_array_equal(got, w)
    ctx.close()


@pytest.mark.parametrize("K", [12, 40, 50, 64, 100, 105])
def test_padded_and_exact_members_stride_and_chunking(K):
    """The padded members (and the exact ones: 50, 100) through the checkpoint / rebuild layout and both beta strides:
    same records as the oracle."""
    pm, bits, folded = _problem(K, S=333, seed=5)
    pairs = O.enumerate_all_pairs(32)[:70]
    pr = np.array(pairs, dtype=np.uint32).view(capi.PAIR_DTYPE).reshape(-1)
    want = O.decode_pairs_ibd(pm, folded, pairs, batch_size=64)
    for stride in (1, 2):
        for chunk in (0, 48):
            ctx = capi.Context(0)
            model = ctx.create_model(pm)
            ctx.upload_haps(bits, pm.S)
            ctx.set_beta_stride(stride)
            if chunk:
                ctx.set_chunk_sites(chunk)
            got = ctx.decode_ibd(model, pr, capi.whole_sequence_groups(len(pairs), pm.S))
            assert ctx.last_beta_stride() == stride
            assert got.size == want.size
            for f_got, f_want in (("pair", "pair"), ("start", "start"), ("end", "end"), ("prob", "prob"),
                                  ("post_mean", "postMean"), ("map", "map")):
                np.testing.assert_array_equal(got[f_got], want[f_want], err_msg=f"{f_got} stride {stride} chunk {chunk}")
            ctx.close()


@pytest.mark.parametrize("K,time", [(256, 9000), (256, 20000), (200, 20000), (150, 20000)])
def test_wide_model_scan_thresholds_that_reach_the_upper_waves(K, time):
    """The IBD scan sums the posterior of the states below the time threshold in state order: with the four-waves-per-
    group kernel that sum starts in wave 0 and continues in as many waves as the threshold reaches (here two, three
    and four of them; the other tests of this file stay inside wave 0)."""
    tables = synth.make_model_tables(K)
    haps = synth.make_haps(64, 200, seed=K + 1, cm_per_mb=25.0, switch_per_cm=0.6)
    bits, derived, flipped = synth.fold_and_pack(haps.alleles)
    folded = np.where(flipped[None, :], 1 - haps.alleles, haps.alleles).astype(np.uint8)
    pm = O.prepare_model(tables, (haps.cm / 100.0).astype(np.float32), haps.bp, derived, 64, time=time)
    kh = 48 if K <= 192 else 64
    assert pm.state_threshold > kh
    pairs = O.enumerate_all_pairs(32)[:80]
    pr = np.array(pairs, dtype=np.uint32).view(capi.PAIR_DTYPE).reshape(-1)
    ctx = capi.Context(0)
    model = ctx.create_model(pm)
    ctx.upload_haps(bits, pm.S)
    want = O.decode_pairs_ibd(pm, folded, pairs, batch_size=64)
    for flags in (capi.FSMC_WANT_MEAN | capi.FSMC_WANT_MAP, 0):
        got = ctx.decode_ibd(model, pr, capi.whole_sequence_groups(len(pairs), pm.S), flags)
        assert ctx.last_kernel() == 1000 + kh
        assert got.size == want.size and got.size > 0
        for f_got, f_want in (("pair", "pair"), ("start", "start"), ("end", "end"), ("prob", "prob")):
            np.testing.assert_array_equal(got[f_got], want[f_want], err_msg=f_got)
        if flags:
            np.testing.assert_array_equal(got["post_mean"], want["postMean"])
            np.testing.assert_array_equal(got["map"], want["map"])
    ctx.close()


def test_too_many_states_is_rejected():
    """Up to 1024 states a kernel holds a pair's vectors in registers; beyond, the any-K kernel keeps them in the workspace
    (tested above: 1025 states); the library's limit is 4096."""
    pm, bits, _ = _problem(16)
    ctx = capi.Context(0)
    import copy
    big = copy.copy(pm)
    big.K = 4100  # every array widened: the shapes are consistent, the C side refuses the size
    pad = lambda a: np.pad(np.asarray(a, np.float32), [(0, 0)] * (np.ndim(a) - 1) + [(0, big.K - pm.K)])  # noqa: E731
    for f in ("pi", "col_ratios", "exp_times", "D", "B", "U", "RR", "e1", "e0m1", "e2m0"):
        setattr(big, f, pad(getattr(pm, f)))
    with pytest.raises(capi.FsmcError):
        ctx.create_model(big)
    ctx.close()


@pytest.mark.parametrize("K", [300, 350, 402, 500, 530, 600, 700, 900, 1030])
def test_beyond_256_states_windows_chunks_and_thresholds(K):
    """Models of more than 256 states -- 300: the wave-group kernel with four waves of 80 states; 350, 402, 500: six,
    seven, eight waves of 64; 530, 600 / 700 / 900: eight waves of 80 / 96 / 128 without landing zones; 1030: the any-K kernel -- through the checkpoint / rebuild layout (explicit chunk lengths that do and do not
    divide the windows), windows whose scan ends before the decode window does, one- and two-site windows, ragged
    groups, with and without segment ages, and a time threshold that puts the scan's state threshold beyond 256 (the
    scan's sum then walks several waves of a group)."""
    tables = synth.make_model_tables(K)
    haps = synth.make_haps(64, 333, seed=K, cm_per_mb=25.0, switch_per_cm=0.6)
    bits, derived, flipped = synth.fold_and_pack(haps.alleles)
    folded = np.where(flipped[None, :], 1 - haps.alleles, haps.alleles).astype(np.uint8)
    gen = (haps.cm / 100.0).astype(np.float32)
    wins = [(0, 64, 0, 333, 0, 333), (64, 40, 3, 330, 3, 330), (104, 64, 10, 331, 37, 300), (168, 9, 100, 101, 100, 101),
            (177, 33, 200, 202, 200, 202), (210, 64, 5, 222, 6, 221), (274, 20, 0, 49, 0, 49), (294, 64, 290, 333, 301, 333),
            (358, 64, 0, 333, 100, 150)]
    n = wins[-1][0] + wins[-1][1]
    pairs = O.enumerate_all_pairs(32)[100:100 + n]
    groups = np.zeros(len(wins), capi.GROUP_DTYPE)
    for g, w in zip(groups, wins):
        g["first_pair"], g["n_pairs"], g["from"], g["to"], g["scan_from"], g["scan_to"] = w
    pr = np.array(pairs, dtype=np.uint32).view(capi.PAIR_DTYPE).reshape(-1)
    for time, flags in ((200, capi.FSMC_WANT_MEAN | capi.FSMC_WANT_MAP), (30000, 0), (30000, capi.FSMC_WANT_MAP)):
        pm = O.prepare_model(tables, gen, haps.bp, derived, 64, time=time)
        if time == 30000:
            assert pm.state_threshold > 256
        want = []
        for first, cnt, frm, to, sfrm, sto in wins:
            sub = pairs[first:first + cnt]
            ob = np.stack([(folded[a] ^ folded[b])[frm:to] for a, b in sub])
            hb = np.stack([(folded[a] & folded[b])[frm:to] for a, b in sub])
            post, _ = O.decode_batch(pm, ob, hb, frm, to)
            full = np.zeros((pm.S, pm.K, cnt), np.float32)
            full[frm:to] = post[frm:to]
            for v in range(cnt):
                want.append(O.ibd_scan_pair(pm, full, v, sfrm, sto, pair_ordinal=first + v,
                                            want_mean=bool(flags & capi.FSMC_WANT_MEAN),
                                            want_map=bool(flags & capi.FSMC_WANT_MAP)))
        want = np.concatenate(want)
        assert want.size > 10
        for chunk in (0, 48, 37):
            ctx = capi.Context(0)
            model = ctx.create_model(pm)
            ctx.upload_haps(bits, pm.S)
            if chunk:
                ctx.set_chunk_sites(chunk)
            got = ctx.decode_ibd(model, pr, groups, flags)
            assert ctx.last_kernel() == _member(K) and (ctx.info()["max_chunks"] > 1) == bool(chunk)
            ctx.close()
            assert got.size == want.size
            for f_got, f_want in (("pair", "pair"), ("start", "start"), ("end", "end"), ("prob", "prob"),
                                  ("post_mean", "postMean"), ("map", "map")):
                np.testing.assert_array_equal(got[f_got], want[f_want], err_msg=f"{f_got} time {time} chunk {chunk}")


def test_mis_shaped_model_arrays_are_rejected_before_the_c_call():
    pm, bits, _ = _problem(16)
    ctx = capi.Context(0)
    import copy
    bad = copy.copy(pm)
    bad.e1 = pm.e1[:-1]  # one site short: fsmc_model_create would read S*K floats
    with pytest.raises(ValueError, match="e1"):
        ctx.create_model(bad)
    bad = copy.copy(pm)
    bad.D = pm.D[:, :-1]
    with pytest.raises(ValueError, match="D"):
        ctx.create_model(bad)
    ctx.close()
