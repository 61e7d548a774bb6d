"""Two half-groups per wavefront (decode_kernel<..., DUAL>; fsmc_ctx_set_pairing): hashing-mode batches of at most 32
pairs share a wave, each lane decoded over its OWN group's decode and scan windows.  The records must be the bytes of
the unpaired run (and of the oracle): different windows in the two halves, a half that starts later / ends earlier than
the other, one-site windows, ragged halves, segment ages on and off, several family members, both beta strides, windows
too long for a wave's workspace (the paired kernel then decodes them in chunks).  Groups that do not pair (more than 32
pairs) run in a second kernel of the same decode and land in the same record list; half-full groups that find no partner
ride in the paired kernel as items of their own."""
import numpy as np
import pytest

from fastsmc_amd import capi, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu
FIELDS = (("pair", "pair"), ("start", "start"), ("end", "end"), ("prob", "prob"), ("post_mean", "postMean"),
          ("map", "map"))


def _oracle(pm, folded, pairs, wins, want_mean, want_map):
    S, out = pm.S, []
    for first, cnt, frm, to, sfrm, sto in wins:
        sub = pairs[first:first + cnt]
        ob = np.stack([(folded[a] ^ folded[b])[frm:to] for a, b in sub])
        hb = np.stack([(folded[a] & folded[b])[frm:to] for a, b in sub])
        post, _ = O.decode_batch(pm, ob, hb, frm, to)
        full = np.zeros((S, pm.K, cnt), np.float32)
        full[frm:to] = post[frm:to]
        for v in range(cnt):
            out.append(O.ibd_scan_pair(pm, full, v, sfrm, sto, want_mean=want_mean, want_map=want_map,
                                       pair_ordinal=first + v))
    return np.concatenate(out)


def _run(ctx, model, pairs, wins, flags, pairing):
    groups = np.zeros(len(wins), capi.GROUP_DTYPE)
    for g, w in zip(groups, wins):
        g["first_pair"], g["n_pairs"], g["from"], g["to"], g["scan_from"], g["scan_to"] = w
    ctx.set_pairing(pairing)
    pr = np.array(pairs, dtype=np.uint32).view(capi.PAIR_DTYPE).reshape(-1)
    rec = ctx.decode_ibd(model, pr, groups, flags)
    return rec, ctx.last_items()


@pytest.mark.parametrize("flags", [capi.FSMC_WANT_MEAN | capi.FSMC_WANT_MAP, 0])
def test_paired_half_groups_match_the_unpaired_run_and_the_oracle(small_problem, flags):
    pm, folded, S = small_problem["model"], small_problem["folded"], small_problem["model"].S
    allp = O.enumerate_all_pairs(32)
    # (count, from, to, scan_from, scan_to): windows chosen so that the sort pairs different ones
    shapes = [(32, 100, 420, 110, 400), (32, 120, 440, 130, 430),      # overlapping, B starts and ends later
              (17, 0, 300, 0, 300), (32, 5, 320, 40, 310),             # ragged half, window at the sequence start
              (32, 300, S, 320, S), (9, 330, S, 330, S - 1),           # windows ending at the last site
              (1, 200, 201, 200, 201), (32, 200, 202, 200, 202),       # one- and two-site windows
              (64, 50, 500, 60, 480), (40, 60, 510, 60, 510),          # full groups: second kernel of the same decode
              (32, 10, 600, 300, 310), (32, 20, 610, 25, 600),         # a short scan window inside a long decode window
              (32, 50, 301, 60, 290), (32, 60, 331, 60, 331)]          # windows that end at an even offset of the union: with
                                                                        # beta stride 2 that row is recomputed in the alpha
                                                                        # sweep, where the lanes of the half that ends there
                                                                        # must start from beta = 1 again
    wins, first = [], 0
    for cnt, frm, to, sf, st in shapes:
        wins.append((first, cnt, frm, to, sf, st))
        first += cnt
    pairs = allp[7:7 + first]
    ctx = capi.Context(0)
    model = ctx.create_model(pm)
    ctx.upload_haps(small_problem["bits"], S)
    plain, n0 = _run(ctx, model, pairs, wins, flags, pairing=0)
    paired, n1 = _run(ctx, model, pairs, wins, flags, pairing=1)
    assert ctx.last_beta_stride() == 2
    ctx.set_beta_stride(1)
    paired1, n2 = _run(ctx, model, pairs, wins, flags, pairing=1)
    assert ctx.last_beta_stride() == 1
    ctx.close()
    assert n0 == 0 and 0 < n1 < len(wins) and n2 > 0  # the paired runs really put two groups on one wave
    assert paired.tobytes() == plain.tobytes() and paired1.tobytes() == plain.tobytes()
    want = _oracle(pm, folded, pairs, wins, bool(flags & capi.FSMC_WANT_MEAN), bool(flags & capi.FSMC_WANT_MAP))
    assert paired.size == want.size and want.size > 10
    for f_got, f_want in FIELDS:
        np.testing.assert_array_equal(paired[f_got], want[f_want], err_msg=f_got)


@pytest.mark.parametrize("K", [12, 50, 100, 105])
def test_pairing_in_the_padded_and_exact_members(K):
    tables = synth.make_model_tables(K)
    haps = synth.make_haps(64, 500, seed=9, cm_per_mb=25.0, switch_per_cm=0.6)
    bits, derived, flipped = synth.fold_and_pack(haps.alleles)
    folded = np.where(flipped[None, :], 1 - haps.alleles, haps.alleles).astype(np.uint8)
    pm = O.prepare_model(tables, (haps.cm / 100.0).astype(np.float32), haps.bp, derived, 64, time=200)
    allp = O.enumerate_all_pairs(32)
    wins = [(0, 32, 50, 350, 60, 340), (32, 30, 70, 380, 70, 380), (62, 32, 200, 500, 210, 500), (94, 5, 190, 480, 190, 470)]
    pairs = allp[:99]
    ctx = capi.Context(0)
    model = ctx.create_model(pm)
    ctx.upload_haps(bits, pm.S)
    plain, _ = _run(ctx, model, pairs, wins, capi.FSMC_WANT_MEAN | capi.FSMC_WANT_MAP, 0)
    paired, n1 = _run(ctx, model, pairs, wins, capi.FSMC_WANT_MEAN | capi.FSMC_WANT_MAP, 1)
    ctx.close()
    assert n1 == 2 and paired.tobytes() == plain.tobytes()
    want = _oracle(pm, folded, pairs, wins, True, True)
    assert paired.size == want.size
    for f_got, f_want in FIELDS:
        np.testing.assert_array_equal(paired[f_got], want[f_want], err_msg=f_got)


def test_long_windows_pair_up_in_the_chunked_layout(small_problem):
    """With a small workspace the long windows do not fit a wave whole: the paired kernel then decodes them in chunks
    (checkpoints, rebuild pass -- the lanes of the half whose window ends inside a chunk start from beta = 1 there in
    the rebuild as in the backward pass), two groups to a wave all the same."""
    pm, folded, S = small_problem["model"], small_problem["folded"], small_problem["model"].S
    allp = O.enumerate_all_pairs(32)
    shapes = [(32, 0, S, 0, S), (32, 100, 180, 100, 180), (32, 110, 190, 110, 190), (20, 5, S - 3, 10, S - 3),
              (32, 400, 470, 400, 470), (32, 402, 480, 402, 480), (32, 0, S, 50, 600), (32, 30, S - 40, 30, S - 41)]
    wins, first = [], 0
    for cnt, frm, to, sf, st in shapes:
        wins.append((first, cnt, frm, to, sf, st))
        first += cnt
    pairs = allp[3:3 + first]
    flags = capi.FSMC_WANT_MEAN | capi.FSMC_WANT_MAP
    ctx = capi.Context(0)
    model = ctx.create_model(pm)
    ctx.upload_haps(small_problem["bits"], S)
    # ~200 rows of 18 float4 x 64 lanes for each wave: with beta stride 2 enough for windows of up to 387 sites whole, not
    # for the 600-site ones
    ctx.set_workspace_limit(200 * 18 * 64 * 16 * len(shapes))
    ctx.set_chunk_sites(96)
    plain, n0 = _run(ctx, model, pairs, wins, flags, pairing=0)
    paired, n1 = _run(ctx, model, pairs, wins, flags, pairing=1)
    chunks = ctx.info()["max_chunks"]
    ctx.close()
    assert n0 == 0 and n1 == 4 and chunks > 1
    assert paired.tobytes() == plain.tobytes()
    want = _oracle(pm, folded, pairs, wins, True, True)
    assert paired.size == want.size
    for f_got, f_want in FIELDS:
        np.testing.assert_array_equal(paired[f_got], want[f_want], err_msg=f_got)
