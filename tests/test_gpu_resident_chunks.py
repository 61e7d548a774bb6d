"""Resident chunks (fsmc_ctx_set_resident_chunks): in a chunked window the backward pass leaves the beta rows of the
window's first chunks in the workspace, and the forward sweep of those chunks skips the rebuild pass.  The rows are the
ones the rebuild would produce -- the same operations on the same operands -- so the records must not depend on how many
chunks are resident: 0, 1, 2, all of them; beta stride 1 and 2; with and without segment ages; windows whose scan ends
inside a chunk; against the oracle bit for bit."""
import numpy as np
import pytest

from fastsmc_amd import capi
from oracle import oracle as O
from test_gpu_beta_stride import FIELDS, _assert_records_equal, _ctx, _pairs_array, _window_oracle

pytestmark = pytest.mark.gpu

WINS = [  # first pair, pairs, from, to, scan_from, scan_to
    (0, 64, 0, 640, 0, 640), (64, 40, 3, 636, 3, 636), (104, 64, 10, 331, 37, 300), (168, 9, 100, 101, 100, 101),
    (177, 33, 200, 202, 200, 202), (210, 64, 300, 303, 301, 303), (274, 64, 5, 422, 6, 421), (338, 20, 0, 49, 0, 49),
    (358, 64, 590, 640, 601, 640), (422, 64, 0, 640, 100, 333),
]


@pytest.mark.parametrize("stride", [1, 2])
@pytest.mark.parametrize("flags", [capi.FSMC_WANT_MEAN | capi.FSMC_WANT_MAP, 0])
def test_records_do_not_depend_on_resident_chunks(small_problem, stride, flags):
    pm = small_problem["model"]
    n = WINS[-1][0] + WINS[-1][1]
    pairs = O.enumerate_all_pairs(32)[500:500 + n]
    groups = np.zeros(len(WINS), capi.GROUP_DTYPE)
    for g, w in zip(groups, WINS):
        g["first_pair"], g["n_pairs"], g["from"], g["to"], g["scan_from"], g["scan_to"] = w
    res = {}
    for resident in (0, 1, 2, -1):
        ctx, model = _ctx(small_problem, stride, limit=(1 << 30) if resident < 0 else (64 << 20), chunk=48)
        ctx.set_resident_chunks(resident)
        res[resident] = ctx.decode_ibd(model, _pairs_array(pairs), groups, flags)
        info = ctx.info()
        assert info["max_chunks"] > 10  # the 640-site windows are chunked
        got_resident = ctx.last_resident_chunks()
        assert got_resident == (resident if resident >= 0 else info["max_chunks"])
        ctx.close()
    for r in (1, 2, -1):
        assert res[r].tobytes() == res[0].tobytes(), f"resident = {r}"
    want = []
    for first, cnt, frm, to, sfrm, sto in WINS:
        full = _window_oracle(small_problem, pairs[first:first + cnt], frm, to, sfrm, sto)
        for v in range(cnt):
            want.append(O.ibd_scan_pair(pm, full, v, sfrm, sto, pair_ordinal=first + v,
                                        want_mean=bool(flags & capi.FSMC_WANT_MEAN),
                                        want_map=bool(flags & capi.FSMC_WANT_MAP)))
    want = np.concatenate(want)
    assert want.size > 20
    _assert_records_equal(res[-1], want)


def test_resident_chunks_setting_is_validated(small_problem):
    ctx = capi.Context(0)
    with pytest.raises(capi.FsmcError):
        ctx.set_resident_chunks(-2)
    ctx.close()


def test_workspace_is_earned_without_a_limit(small_problem, monkeypatch):
    """Without a caller's limit a plan is upgraded (here: resident chunks) only with memory the context's launches have
    paid for (hipMalloc costs 40 ms per GB).  Made visible on a small problem by two diagnostic
    switches: the free allowance down to one byte, and an earning rate that turns this launch into a long job's."""
    n = WINS[-1][0] + WINS[-1][1]
    pairs = O.enumerate_all_pairs(32)[500:500 + n]
    groups = np.zeros(len(WINS), capi.GROUP_DTYPE)
    for g, w in zip(groups, WINS):
        g["first_pair"], g["n_pairs"], g["from"], g["to"], g["scan_from"], g["scan_to"] = w
    flags = capi.FSMC_WANT_MEAN | capi.FSMC_WANT_MAP

    def run(launches):
        ctx, model = _ctx(small_problem, 2, limit=0, chunk=48)
        out = []
        for _ in range(launches):
            rec = ctx.decode_ibd(model, _pairs_array(pairs), groups, flags)
            out.append((ctx.last_resident_chunks(), rec.tobytes()))
        chunks = ctx.info()["max_chunks"]
        ctx.close()
        return out, chunks

    monkeypatch.setenv("FSMC_DIAG_WS_FREE", "1")
    poor, chunks = run(3)
    assert chunks > 10 and [r for r, _ in poor] == [0, 0, 0]     # nothing earned to speak of: every chunk rebuilt
    monkeypatch.setenv("FSMC_DIAG_WS_EARN_SCALE", "1e9")
    rich, _ = run(2)
    assert rich[0][0] == chunks and rich[1][0] == chunks          # the launch itself pays for its upgrade
    monkeypatch.delenv("FSMC_DIAG_WS_FREE")
    monkeypatch.delenv("FSMC_DIAG_WS_EARN_SCALE")
    free, _ = run(1)
    assert free[0][0] == chunks                                   # a problem this small fits the free allowance
    assert len({b for _, b in poor + rich + free}) == 1           # the records do not depend on the plan


def _groups_and_pairs():
    n = WINS[-1][0] + WINS[-1][1]
    pairs = O.enumerate_all_pairs(32)[500:500 + n]
    groups = np.zeros(len(WINS), capi.GROUP_DTYPE)
    for g, w in zip(groups, WINS):
        g["first_pair"], g["n_pairs"], g["from"], g["to"], g["scan_from"], g["scan_to"] = w
    return pairs, groups


def test_an_announced_job_has_its_workspace_at_the_first_launch(small_problem, monkeypatch):
    """fsmc_ctx_expect_work (what HMM::decodeAll knows when it starts, HMM.cpp:310-321): under the library's own policy
    the credit of the whole announced job is there at the first launch -- here a job announced big enough pays for
    every chunk's rows at once, with the free allowance switched off -- and the records do not depend on it."""
    pairs, groups = _groups_and_pairs()
    flags = capi.FSMC_WANT_MEAN | capi.FSMC_WANT_MAP
    monkeypatch.setenv("FSMC_DIAG_WS_FREE", "1")
    out = {}
    for announce in (0.0, 1e15):
        ctx, model = _ctx(small_problem, 2, limit=0, chunk=48)
        if announce:
            ctx.expect_work(announce, small_problem["model"].K)
        rec = ctx.decode_ibd(model, _pairs_array(pairs), groups, flags)
        out[announce] = (ctx.last_resident_chunks(), ctx.info()["max_chunks"], rec.tobytes())
        ctx.close()
    assert out[0.0][0] == 0 and out[1e15][0] == out[1e15][1] > 10
    assert out[0.0][2] == out[1e15][2]
    # the announced job ends before it launches (pair_sites = 0: HMM::finishDecoding): the unspent credit is taken back
    # and nothing of the announcement stays behind -- the next launch is a young context's again
    ctx, model = _ctx(small_problem, 2, limit=0, chunk=48)
    ctx.expect_work(1e15, small_problem["model"].K)
    ctx.expect_work(0.0, small_problem["model"].K)
    rec = ctx.decode_ibd(model, _pairs_array(pairs), groups, flags)
    assert ctx.last_resident_chunks() == 0 and rec.tobytes() == out[0.0][2]
    ctx.close()
    ctx, _ = _ctx(small_problem, 2, limit=0, chunk=48)
    with pytest.raises(capi.FsmcError):
        ctx.expect_work(-1.0, 69)
    ctx.close()


def test_workspace_growth_is_amortised(small_problem, monkeypatch):
    """A bigger workspace is a new allocation of its whole size (hipMalloc: ~40 ms per GB), so the plan is upgraded only
    when the credit covers TWICE the buffer held, and an allocation is debited from the credit: with launches that each
    earn a third of the first plan's buffer the number of resident chunks does not creep up launch by launch (what it
    did until round 4: a re-allocation at almost every flush of a long job) -- it stays put for several launches and
    then moves in a few large steps."""
    pm = small_problem["model"]
    pairs, groups = _groups_and_pairs()
    flags = capi.FSMC_WANT_MEAN | capi.FSMC_WANT_MAP
    monkeypatch.setenv("FSMC_DIAG_WS_FREE", "1")
    # the buffer of the plan without upgrades: (chunk rows + checkpoints + 4 side rows) of K/4 x 64 float4 per slot
    ctx, model = _ctx(small_problem, 2, limit=0, chunk=48)
    ctx.decode_ibd(model, _pairs_array(pairs), groups, flags)
    info = ctx.info()
    assert ctx.last_resident_chunks() == 0
    ctx.close()
    k4 = (pm.K + 3) // 4
    slots = min(len(WINS), info["n_slots"])
    base_bytes = (48 // 2 + info["max_chunks"] + 4) * k4 * 64 * 16 * slots
    pair_sites = float(sum(w[1] * (w[5] - w[2]) for w in WINS))
    seconds = pair_sites * (8 * pm.K + 0.25) / (0.8 * 8e12)
    per_launch = base_bytes / 3.0  # what one launch is to earn
    monkeypatch.setenv("FSMC_DIAG_WS_EARN_SCALE", repr(per_launch / (0.06 * seconds * 25e9)))
    ctx, model = _ctx(small_problem, 2, limit=0, chunk=48)
    resident, recs = [], set()
    for _ in range(24):
        recs.add(ctx.decode_ibd(model, _pairs_array(pairs), groups, flags).tobytes())
        resident.append(ctx.last_resident_chunks())
    ctx.close()
    assert len(recs) == 1
    assert resident == sorted(resident) and resident[-1] > 0          # it does grow ...
    assert resident[:4] == [0, 0, 0, 0]                                  # ... not before the credit covers 2 x the buffer
    steps = sum(1 for a, b in zip(resident, resident[1:]) if b != a)
    assert steps <= 4, resident                                          # ... and in a few steps, not one per launch


@pytest.mark.parametrize("K", [200, 300, 530])
def test_wave_group_kernel_records_and_sums_do_not_depend_on_resident_chunks(K):
    """The wave-group kernel's four-wave members (here 64 and 80 states a wave) keep resident chunks too: records with ages
    and the sums over pairs of chunked windows are the same bytes with 0, 1, 2 and every chunk resident, and the records
    are the oracle's.  The members of more waves (530 states: eight waves of 80) are not built with them: the setting is
    accepted and the plan has none."""
    from fastsmc_amd import synth
    from conftest import expected_member

    tables = synth.make_model_tables(K)
    haps = synth.make_haps(64, 333, seed=K, cm_per_mb=25.0, switch_per_cm=0.6)
    bits, derived, flipped = synth.fold_and_pack(haps.alleles)
    folded = np.where(flipped[None, :], 1 - haps.alleles, haps.alleles).astype(np.uint8)
    gen = (haps.cm / 100.0).astype(np.float32)
    pm = O.prepare_model(tables, gen, haps.bp, derived, 64, time=200)
    wins = [(0, 64, 0, 333, 0, 333), (64, 40, 3, 330, 3, 330), (104, 64, 10, 331, 37, 300), (168, 9, 100, 101, 100, 101),
            (177, 33, 200, 202, 200, 202), (210, 64, 5, 222, 6, 221), (274, 20, 0, 49, 0, 49), (294, 64, 290, 333, 301, 333),
            (358, 64, 0, 333, 100, 150)]
    n = wins[-1][0] + wins[-1][1]
    pairs = O.enumerate_all_pairs(32)[100:100 + n]
    groups = np.zeros(len(wins), capi.GROUP_DTYPE)
    for g, w in zip(groups, wins):
        g["first_pair"], g["n_pairs"], g["from"], g["to"], g["scan_from"], g["scan_to"] = w
    pr = np.array(pairs, dtype=np.uint32).view(capi.PAIR_DTYPE).reshape(-1)
    flags = capi.FSMC_WANT_MEAN | capi.FSMC_WANT_MAP
    rec, sums = {}, {}
    for resident in (0, 1, 2, -1):
        ctx = capi.Context(0)
        model = ctx.create_model(pm)
        ctx.upload_haps(bits, pm.S)
        ctx.set_chunk_sites(48)
        ctx.set_workspace_limit(1 << 30)
        ctx.set_resident_chunks(resident)
        rec[resident] = ctx.decode_ibd(model, pr, groups, flags)
        assert ctx.last_kernel() == expected_member(K)
        chunks = ctx.info()["max_chunks"]
        assert chunks > 4
        built = expected_member(K) < 2000  # (1000 + states per wave: four waves a group)
        assert ctx.last_resident_chunks() == ((resident if resident >= 0 else chunks) if built else 0)
        ctx.upload_worklist(pr[:64], capi.whole_sequence_groups(64, pm.S))
        sums[resident] = ctx.decode_sums(model)[0]
        assert ctx.last_resident_chunks() == ((resident if resident >= 0 else ctx.info()["max_chunks"]) if built else 0)
        ctx.close()
    for r in (1, 2, -1):
        assert rec[r].tobytes() == rec[0].tobytes(), f"records, resident = {r}"
        assert sums[r].tobytes() == sums[0].tobytes(), f"sums, resident = {r}"
    want = []
    for first, cnt, frm, to, sfrm, sto in wins:
        sub = pairs[first:first + cnt]
        ob = np.stack([(folded[a] ^ folded[b])[frm:to] for a, b in sub])
        hb = np.stack([(folded[a] & folded[b])[frm:to] for a, b in sub])
        post, _ = O.decode_batch(pm, ob, hb, frm, to)
        full = np.zeros((pm.S, pm.K, cnt), np.float32)
        full[frm:to] = post[frm:to]
        for v in range(cnt):
            want.append(O.ibd_scan_pair(pm, full, v, sfrm, sto, pair_ordinal=first + v, want_mean=True, want_map=True))
    want = np.concatenate(want)
    assert want.size > 10
    _assert_records_equal(rec[-1], want)
    ob = np.stack([folded[a] ^ folded[b] for a, b in pairs[:64]])
    hb = np.stack([folded[a] & folded[b] for a, b in pairs[:64]])
    wpost, _ = O.decode_batch(pm, ob, hb, 0, pm.S)
    wsum = np.zeros((pm.S, pm.K), np.float32)
    O.augment_sum_over_pairs(pm, wpost, 64, ob, hb, wsum)
    np.testing.assert_array_equal(sums[-1], wsum)
