"""The other posterior consumers on the GPU: per-pair posterior mean / MAP (HMM::writePerPairOutput,
HMM.cpp:1360-1458), sums over pairs (HMM::augmentSumOverPairs, HMM.cpp:1044-1085) and the ASMC pair-list API
(ASMC.cpp:80-128), against the oracle."""
import copy

import numpy as np
import pytest

from fastsmc_amd import api, capi, synth
from oracle import oracle as O

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("window_waves")]


@pytest.fixture(scope="module")
def gpu(small_problem):
    ctx = capi.Context(0)
    model = ctx.create_model(small_problem["model"])
    ctx.upload_haps(small_problem["bits"], small_problem["model"].S)
    yield ctx, model
    ctx.close()


def _pairs_array(pairs):
    return np.array(pairs, dtype=np.uint32).view(capi.PAIR_DTYPE).reshape(-1)


def _oracle_batches(sp, pairs, batch):
    folded, pm = sp["folded"], sp["model"]
    for b0 in range(0, len(pairs), batch):
        chunk = pairs[b0:b0 + batch]
        ob = np.stack([folded[a] ^ folded[b] for a, b in chunk])
        hb = np.stack([folded[a] & folded[b] for a, b in chunk])
        post, _ = O.decode_batch(pm, ob, hb, 0, pm.S)
        yield b0, chunk, ob, hb, post


def test_per_pair_mean_and_map_bit_exact(gpu, small_problem):
    ctx, model = gpu
    pm = small_problem["model"]
    pairs = O.enumerate_all_pairs(32)[100:100 + 150]
    ctx.upload_worklist(_pairs_array(pairs), capi.whole_sequence_groups(len(pairs), pm.S))
    mean, mp = ctx.decode_per_pair(model, pm.exp_times)
    for b0, chunk, _, _, post in _oracle_batches(small_problem, pairs, 64):
        wmean, wmap, _ = O.per_pair_output(pm, post, len(chunk))
        np.testing.assert_array_equal(mean[b0:b0 + len(chunk)], wmean)
        np.testing.assert_array_equal(mp[b0:b0 + len(chunk)], wmap)


@pytest.mark.parametrize("batch", [64, 32])
def test_sum_over_pairs(gpu, small_problem, batch):
    """With one group per resident wave the additions happen in the reference's order: bit-exact."""
    ctx, model = gpu
    pm = small_problem["model"]
    pairs = O.enumerate_all_pairs(32)[:200]
    ctx.upload_worklist(_pairs_array(pairs), capi.whole_sequence_groups(len(pairs), pm.S, batch=batch))
    s, (s00, s01, s11) = ctx.decode_sums(model, major_minor=True)
    want = [np.zeros((pm.S, pm.K), np.float32) for _ in range(4)]
    for _, chunk, ob, hb, post in _oracle_batches(small_problem, pairs, batch):
        O.augment_sum_over_pairs(pm, post, len(chunk), ob, hb, want[0], want[1], want[2], want[3])
    for got, w in zip((s, s00, s01, s11), want):
        np.testing.assert_array_equal(got, w)
    np.testing.assert_allclose(s, s00 + s01 + s11, rtol=1e-5)
    np.testing.assert_allclose(s.sum(axis=1), len(pairs), rtol=1e-5)


def _write_files(sp, root):
    synth.write_haps_files(root, sp["haps"], fastsmc_map=False)  # plink map for ASMC mode (Data.cpp:162-210)
    t = copy.copy(sp["tables"])
    synth.write_decoding_quantities(root + ".decodingQuantities.gz", _subset_keys(t, sp))


def _subset_keys(t, sp):
    # ASMC mode computes gen = stof(cM)/100.f in float (Data.cpp:186): keep every key to be safe but small
    gen = np.array([np.float32(np.float32(c) / np.float32(100.0)) for c in sp["haps"].cm], np.float32)
    used = np.unique(np.concatenate([[0.0], O.step_rows(t.keys, gen)[1][1:], O.step_rows(t.keys, sp["gen"])[1][1:]]))
    sel = np.nonzero(np.isin(t.keys, used.astype(np.float32)))[0]
    t.keys, t.D, t.B, t.U, t.RR = t.keys[sel], t.D[sel], t.B[sel], t.U[sel], t.RR[sel]
    return t


def test_asmc_decode_pairs_api(small_problem, tmp_path):
    sp = small_problem
    root = str(tmp_path / "asmc")
    _write_files(sp, root)
    # the (in_dir, dq_file) constructor seeds the emission RNG from std::random_device like the reference
    # (ASMC.cpp:28-49, Data.cpp:55-60); for a reproducible comparison build the same params with the known seed
    p0 = api.DecodingParams(root, root + ".decodingQuantities.gz", root, 1, 1, "array", False, True, False, False,
                            0.0, False, True, False, "", False, True)
    p0.doPerPairMAP = True
    p0.useKnownSeed = True
    asmc = api.ASMC(p0)
    a = [1, 2, 3, 10, 40, 63, 7]
    b = [2, 3, 4, 11, 41, 0, 9]
    asmc.decodePairs(a, b, True, True, True, True)
    res = asmc.get_copy_of_results()
    # oracle on the same data as the ASMC-mode readers see it
    p = api.DecodingParams(root, root + ".decodingQuantities.gz")
    p.useKnownSeed = True
    data = api.Data(p)
    gen = np.array(data.geneticPositions, np.float32)
    _, derived, _ = synth.fold_and_pack(sp["haps"].alleles)
    pm = O.prepare_model(sp["tables"], gen, sp["haps"].bp, derived, 64, time=p.time, no_conditional_age_estimates=False)
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
    np.testing.assert_array_equal(res.min_posterior_means, wmean.min(axis=0))
    np.testing.assert_array_equal(res.argmin_posterior_means, wmean.argmin(axis=0))
    np.testing.assert_array_equal(res.min_MAPs, wmap.min(axis=0))
    assert res.per_pair_indices[0] == (1, "1_1#2", 2, "1_2#1")
    # by id string (ASMC.cpp:102-128)
    asmc.decodePairs(["1_1#2", "1_2#1"], ["1_2#1", "1_2#2"], False, False, True, True)
    res2 = asmc.get_copy_of_results()
    np.testing.assert_array_equal(res2.per_pair_posterior_means[:2], wmean[:2])
    with pytest.raises(RuntimeError, match="must be the same size"):
        asmc.decodePairs([1, 2], [3], False, False, True, True)


def test_hmm_decode_all_posterior_sums(small_problem, tmp_path):
    """The reference's ASMC regression shape (test_regression.cpp:23-68): decodeAll + sumOverPairs."""
    sp = small_problem
    root = str(tmp_path / "asmc2")
    _write_files(sp, root)
    p = api.DecodingParams(root, root + ".decodingQuantFile.missing", doPosteriorSums=True)
    p.decodingQuantFile = root + ".decodingQuantities.gz"
    p.useKnownSeed = True
    p.jobs, p.jobInd = 4, 2
    data = api.Data(p)
    hmm = api.HMM(data, p)
    hmm.decodeAll(p.jobs, p.jobInd)
    got = hmm.getDecodingReturnValues().sumOverPairs
    gen = np.array(data.geneticPositions, np.float32)
    _, derived, _ = synth.fold_and_pack(sp["haps"].alleles)
    pm = O.prepare_model(sp["tables"], gen, sp["haps"].bp, derived, 64, time=p.time, no_conditional_age_estimates=False)
    pairs = O.enumerate_all_pairs(32, 4, 2)
    want = np.zeros((pm.S, pm.K), np.float32)
    folded = sp["folded"]
    for b0 in range(0, len(pairs), 64):
        chunk = pairs[b0:b0 + 64]
        ob = np.stack([folded[x] ^ folded[y] for x, y in chunk])
        hb = np.stack([folded[x] & folded[y] for x, y in chunk])
        post, _ = O.decode_batch(pm, ob, hb, 0, pm.S)
        O.augment_sum_over_pairs(pm, post, len(chunk), ob, hb, want)
    assert got.shape == (pm.S, pm.K)
    np.testing.assert_array_equal(got, want)


def test_hmm_posterior_sums_with_a_batch_size_of_128(small_problem, tmp_path):
    """batchSize = 128 through the host API: two groups per batch on the device, one running sum per batch
    (HMM.cpp:1054-1073)."""
    sp = small_problem
    root = str(tmp_path / "asmc3")
    _write_files(sp, root)
    p = api.DecodingParams(root, root + ".decodingQuantFile.missing", doPosteriorSums=True)
    p.decodingQuantFile = root + ".decodingQuantities.gz"
    p.useKnownSeed = True
    p.batchSize = 128
    data = api.Data(p)
    hmm = api.HMM(data, p)
    hmm.decodeAll(1, 1)
    got = hmm.getDecodingReturnValues().sumOverPairs
    gen = np.array(data.geneticPositions, np.float32)
    _, derived, _ = synth.fold_and_pack(sp["haps"].alleles)
    pm = O.prepare_model(sp["tables"], gen, sp["haps"].bp, derived, 64, time=p.time, no_conditional_age_estimates=False)
    pairs = O.enumerate_all_pairs(32)
    want = np.zeros((pm.S, pm.K), np.float32)
    folded = sp["folded"]
    for b0 in range(0, len(pairs), 128):
        chunk = pairs[b0:b0 + 128]
        ob = np.stack([folded[x] ^ folded[y] for x, y in chunk])
        hb = np.stack([folded[x] & folded[y] for x, y in chunk])
        post, _ = O.decode_batch(pm, ob, hb, 0, pm.S)
        O.augment_sum_over_pairs(pm, post, len(chunk), ob, hb, want)
    np.testing.assert_array_equal(got, want)
