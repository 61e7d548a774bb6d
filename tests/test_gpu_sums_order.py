"""Sums of posteriors over pairs (HMM::augmentSumOverPairs, HMM.cpp:1044-1085) in the regimes where the ORDER of the
fp32 additions matters: many more batches than one launch holds (several launches, each adding its batches' sums to
the accumulator one after the other), batches of the reference's default size 32, the 00 / 01 / 11 split, and an
accumulation continued over two calls (two flushes of the host's work list).  The reference adds batch by batch --
sumOverPairs(pos, k) += sum_v posterior -- and so does the device path: bit-identical, not merely within 1e-5."""
import numpy as np
import pytest

from fastsmc_amd import capi, synth
from oracle import oracle as O

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("window_waves")]


def _problem(K=69, n_hap=96, S=1500, seed=3):
    tables = synth.make_model_tables(K)
    haps = synth.make_haps(n_hap, S, seed=seed, cm_per_mb=25.0, switch_per_cm=0.6)
    bits, derived, flipped = synth.fold_and_pack(haps.alleles)
    folded = np.where(flipped[None, :], 1 - haps.alleles, haps.alleles).astype(np.uint8)
    gen = (haps.cm / 100.0).astype(np.float32)
    pm = O.prepare_model(tables, gen, haps.bp, derived, n_hap, time=50)
    return pm, bits, folded


def _oracle_sums(pm, folded, pairs, batch, acc=None):
    """The reference's loop: decode a batch, add its per-site per-state sums (and the split by genotype class)."""
    if acc is None:
        acc = [np.zeros((pm.S, pm.K), np.float32) for _ in range(4)]
    for lo in range(0, len(pairs), batch):
        chunk = pairs[lo:lo + batch]
        ob = np.stack([folded[a] ^ folded[b] for a, b in chunk])
        hb = np.stack([folded[a] & folded[b] for a, b in chunk])
        post, _ = O.decode_batch(pm, ob, hb, 0, pm.S)
        O.augment_sum_over_pairs(pm, post, len(chunk), ob, hb, acc[0], acc[1], acc[2], acc[3])
    return acc


@pytest.mark.parametrize("K", [69, 40])
def test_sums_are_bit_identical_with_many_more_batches_than_a_launch_holds(K, monkeypatch):
    pm, bits, folded = _problem(K=K, n_hap=64 if K != 69 else 96, S=600 if K != 69 else 1500)
    n_ind = folded.shape[0] // 2
    pairs = O.enumerate_all_pairs(n_ind)
    pairs = pairs[: (len(pairs) // 32) * 32 + 7]  # a ragged last batch
    want = _oracle_sums(pm, folded, pairs, 32)
    ctx = capi.Context(0)
    model = ctx.create_model(pm)
    ctx.upload_haps(bits, pm.S)
    pr = np.array(pairs, dtype=np.uint32).view(capi.PAIR_DTYPE).reshape(-1)
    ctx.upload_worklist(pr, capi.whole_sequence_groups(len(pairs), pm.S, batch=32))
    # five groups per launch: (number of batches) / 5 launches, each followed by the ordered add of its planes
    monkeypatch.setenv("FSMC_DIAG_SUMS_SLOTS", "5")
    s, mm = ctx.decode_sums(model, major_minor=True)
    np.testing.assert_array_equal(s, want[0])
    for got, w in zip(mm, want[1:]):
        np.testing.assert_array_equal(got, w)
    ctx.close()


def test_sums_continue_over_two_calls_like_two_flushes():
    pm, bits, folded = _problem(n_hap=64, S=500)
    pairs = O.enumerate_all_pairs(32)[:800]
    first, second = pairs[:320], pairs[320:]
    want = _oracle_sums(pm, folded, first, 32)
    want = _oracle_sums(pm, folded, second, 32, want)
    ctx = capi.Context(0)
    model = ctx.create_model(pm)
    ctx.upload_haps(bits, pm.S)
    acc = None
    for part in (first, second):
        pr = np.array(part, dtype=np.uint32).view(capi.PAIR_DTYPE).reshape(-1)
        ctx.upload_worklist(pr, capi.whole_sequence_groups(len(part), pm.S, batch=32))
        acc = ctx.decode_sums(model, major_minor=True, into=acc)
    np.testing.assert_array_equal(acc[0], want[0])
    for got, w in zip(acc[1], want[1:]):
        np.testing.assert_array_equal(got, w)
    ctx.close()


@pytest.mark.parametrize("K,batch", [(69, 128), (69, 200), (150, 136)])
def test_sums_of_batches_larger_than_a_wavefront(K, batch, monkeypatch):
    """batchSize only has to be a multiple of 8 (DecodingParams.cpp:301): the reference sums a WHOLE batch over its
    pairs in order and then adds it (HMM.cpp:1054-1073).  A batch of more than 64 pairs is several groups here; they
    share one running sum (fsmc_decode_sums_batches), so the result is still the reference's bit for bit -- also when a
    launch holds fewer batches than there are and with a ragged last batch."""
    pm, bits, folded = _problem(K=K, n_hap=64, S=400, seed=11)
    pairs = O.enumerate_all_pairs(32)
    pairs = pairs[: 3 * batch + 77]
    want = _oracle_sums(pm, folded, pairs, batch)
    ctx = capi.Context(0)
    model = ctx.create_model(pm)
    ctx.upload_haps(bits, pm.S)
    pr = np.array(pairs, dtype=np.uint32).view(capi.PAIR_DTYPE).reshape(-1)
    groups, first = [], []
    for lo in range(0, len(pairs), batch):
        n = min(batch, len(pairs) - lo)
        first.append(len(groups))
        for off in range(0, n, 64):
            groups.append((lo + off, min(64, n - off), 0, pm.S, 0, pm.S))
    first.append(len(groups))
    ctx.upload_worklist(pr, np.array(groups, dtype=capi.GROUP_DTYPE))
    monkeypatch.setenv("FSMC_DIAG_SUMS_SLOTS", "2")
    s, mm = ctx.decode_sums(model, major_minor=True, batch_first_group=first)
    np.testing.assert_array_equal(s, want[0])
    for got, w in zip(mm, want[1:]):
        np.testing.assert_array_equal(got, w)
    # the batches must partition the group list
    with pytest.raises(RuntimeError):
        ctx.decode_sums(model, batch_first_group=first[:-1])
    ctx.close()
