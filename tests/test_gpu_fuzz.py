"""Seeded random work lists against the oracle, bit for bit: random group sizes (1..64 pairs), random decode windows,
random scan windows inside them, random beta stride, chunk size and workspace limit, ages on or off.  Each case
checks every field of every IBD record of every pair."""
import numpy as np
import pytest

from fastsmc_amd import capi
from oracle import oracle as O

pytestmark = pytest.mark.gpu

FIELDS = (("pair", "pair"), ("start", "start"), ("end", "end"), ("prob", "prob"), ("post_mean", "postMean"),
          ("map", "map"))


def _other_model(small_problem, K, time):
    """The small problem's haplotypes with a synthetic model of K states (another member of the kernel family)."""
    from fastsmc_amd import synth

    sp = small_problem
    _, derived, _ = synth.fold_and_pack(sp["haps"].alleles)
    return O.prepare_model(synth.make_model_tables(K), sp["gen"], sp["haps"].bp, derived, 64, time=time)


# (K, time threshold): the 69-state exact member; the exact 50- and 100-state members; a padded member; the wave-group
# kernel with two workgroups per CU (200 states, a threshold that reaches its second wave) and with one (300 states: four
# waves of 80; 402: seven waves of 64, a threshold beyond 256 states; 460: eight waves; 530:
# eight waves of 80, no landing zones); the any-K kernel (1030)
MODELS = [(69, 50), (50, 50), (100, 200), (105, 200), (200, 20000), (300, 200), (402, 30000), (460, 200), (530, 200),
          (1030, 200)]


@pytest.mark.parametrize("seed", list(range(10)))
def test_random_worklists(small_problem, seed):
    rng = np.random.default_rng(1000 + seed)
    K, time = MODELS[seed % len(MODELS)] if seed >= 2 else MODELS[0]
    pm = small_problem["model"] if K == 69 else _other_model(small_problem, K, time)
    folded = small_problem["folded"]
    S = pm.S
    allp = O.enumerate_all_pairs(32)
    n_groups = int(rng.integers(2, 7))
    wins, first = [], 0
    for _ in range(n_groups):
        cnt = int(rng.choice([1, 2, 15, 16, 17, 31, 33, 63, 64]))
        frm = int(rng.integers(0, S - 2))
        to = int(rng.integers(frm + 1, min(S, frm + int(rng.choice([1, 2, 3, 40, 200, S]))) + 1))
        sfrm = int(rng.integers(frm, to))
        sto = int(rng.integers(sfrm + 1, to + 1))
        wins.append((first, cnt, frm, to, sfrm, sto))
        first += cnt
    start = int(rng.integers(0, len(allp) - first))
    pairs = allp[start:start + first]
    groups = np.zeros(len(wins), capi.GROUP_DTYPE)
    for g, w in zip(groups, wins):
        g["first_pair"], g["n_pairs"], g["from"], g["to"], g["scan_from"], g["scan_to"] = w
    want_mean, want_map = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    flags = (capi.FSMC_WANT_MEAN if want_mean else 0) | (capi.FSMC_WANT_MAP if want_map else 0)

    ctx = capi.Context(0)
    ctx.set_beta_stride(int(rng.choice([0, 1, 2])))
    if rng.integers(0, 2):
        ctx.set_workspace_limit((int(rng.choice([12, 24, 64])) << 20) * max(1, (pm.K + 63) // 64))
        ctx.set_chunk_sites(int(rng.choice([0, 16, 32, 80])))
    model = ctx.create_model(pm)
    ctx.upload_haps(small_problem["bits"], S)
    pr = np.array(pairs, dtype=np.uint32).view(capi.PAIR_DTYPE).reshape(-1)
    try:
        got = ctx.decode_ibd(model, pr, groups, flags)
    except capi.FsmcError as e:  # a random workspace limit may be too small for the longest window: not a parity case
        ctx.close()
        assert "workspace limit" in str(e)
        pytest.skip(str(e))
    ctx.close()

    want = []
    for first, cnt, frm, to, sfrm, sto in wins:
        sub = pairs[first:first + cnt]
        ob = np.stack([(folded[a] ^ folded[b])[frm:to] for a, b in sub])
        hb = np.stack([(folded[a] & folded[b])[frm:to] for a, b in sub])
        post, _ = O.decode_batch(pm, ob, hb, frm, to)
        full = np.zeros((S, pm.K, cnt), np.float32)
        full[frm:to] = post[frm:to]
        for v in range(cnt):
            want.append(O.ibd_scan_pair(pm, full, v, sfrm, sto, want_mean=want_mean, want_map=want_map,
                                        pair_ordinal=first + v))
    want = np.concatenate(want) if want else np.zeros(0, O.IBD_DTYPE)
    assert got.size == want.size
    for f_got, f_want in FIELDS:
        np.testing.assert_array_equal(got[f_got], want[f_want], err_msg=f_got)


@pytest.mark.parametrize("seed", list(range(8)))
def test_random_hashing_style_worklists_pair_up(small_problem, seed):
    """Half-full groups with similar windows (what the hashing pre-filter produces): most of them are decoded two to a
    wave, at either beta stride, beside the ones that find no partner -- every record against the oracle."""
    rng = np.random.default_rng(7000 + seed)
    pm = small_problem["model"]
    folded = small_problem["folded"]
    S = pm.S
    allp = O.enumerate_all_pairs(32)
    wins, first = [], 0
    for _ in range(int(rng.integers(8, 15))):
        cnt = int(rng.choice([1, 7, 16, 31, 32, 32, 32, 40]))
        length = int(rng.choice([1, 2, 65, 128, 129, 200, 320]))
        frm = int(rng.integers(0, max(1, min(40, S - length))))
        frm += int(rng.choice([0, 0, 100, 250]))
        frm = min(frm, S - length)
        to = frm + length
        sfrm = int(rng.integers(frm, to))
        sto = int(rng.integers(sfrm + 1, to + 1))
        wins.append((first, cnt, frm, to, sfrm, sto))
        first += cnt
    start = int(rng.integers(0, len(allp) - first))
    pairs = allp[start:start + first]
    groups = np.zeros(len(wins), capi.GROUP_DTYPE)
    for g, w in zip(groups, wins):
        g["first_pair"], g["n_pairs"], g["from"], g["to"], g["scan_from"], g["scan_to"] = w
    want_mean, want_map = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    flags = (capi.FSMC_WANT_MEAN if want_mean else 0) | (capi.FSMC_WANT_MAP if want_map else 0)
    ctx = capi.Context(0)
    ctx.set_beta_stride(int(rng.choice([0, 1, 2])))
    model = ctx.create_model(pm)
    ctx.upload_haps(small_problem["bits"], S)
    pr = np.array(pairs, dtype=np.uint32).view(capi.PAIR_DTYPE).reshape(-1)
    got = ctx.decode_ibd(model, pr, groups, flags)
    n_items = ctx.last_items()
    ctx.close()
    assert n_items >= 1  # at least one wave took two groups
    want = []
    for first, cnt, frm, to, sfrm, sto in wins:
        sub = pairs[first:first + cnt]
        ob = np.stack([(folded[a] ^ folded[b])[frm:to] for a, b in sub])
        hb = np.stack([(folded[a] & folded[b])[frm:to] for a, b in sub])
        post, _ = O.decode_batch(pm, ob, hb, frm, to)
        full = np.zeros((S, pm.K, cnt), np.float32)
        full[frm:to] = post[frm:to]
        for v in range(cnt):
            want.append(O.ibd_scan_pair(pm, full, v, sfrm, sto, want_mean=want_mean, want_map=want_map,
                                        pair_ordinal=first + v))
    want = np.concatenate(want) if want else np.zeros(0, O.IBD_DTYPE)
    assert got.size == want.size
    for f_got, f_want in FIELDS:
        np.testing.assert_array_equal(got[f_got], want[f_want], err_msg=f_got)
