"""Known answers the reference's own tests hold for the boundary of the decode path, restated against this build.

* ``TESTS/unit_tests.cpp:26-66``  -- ``StringUtils::stof / stod`` (denormal input, NAN, INF, out_of_range,
  invalid_argument): the parsers of every number in the decoding-quantities and map files.
* ``TESTS/test_HMM.cpp:44-79``    -- batch-buffer bookkeeping of ``HMM::decodePair(s)`` / ``finishDecoding``.
  The reference test runs on the n300 array example with the 30-100-2000 model (a blob missing from the checkout);
  the bookkeeping does not depend on the model, so a synthetic one stands in.
* ``TESTS/test_HMM.cpp:35-42``    -- ``decodeSummarize`` returns MAP and mean rows of the posterior's length (GPU).
* ``Individual`` (``Individual.cpp:18-32``, ``pybind.cpp:89-95``) and the ``asmc`` package names (``__init__.py:18-33``).
"""
import math

import numpy as np
import pytest

from fastsmc_amd import _pyasmc as M
from fastsmc_amd import api


def _hmm(small_problem, batch=64, **kw):
    sp = small_problem
    data = api.Data.from_arrays(sp["haps"].alleles, sp["haps"].bp, sp["haps"].cm, True, True)
    dq = api.decoding_quantities_from_tables(sp["tables"])
    p = api.DecodingParams()
    p.foldData = True
    p.usingCSFS = True
    p.batchSize = batch  # the reference constructor's default (DecodingParams.cpp:38), "default batch size is 64"
    p.useKnownSeed = True
    for k, v in kw.items():
        setattr(p, k, v)
    return api.HMM(data, dq, p), data


def test_stof_stod_known_answers():
    # unit_tests.cpp:47-65
    too_small = "3.20676899524985E-310"  # not representable as a normal float or double
    assert M.stof("1.0") == 1.0 and M.stof("1.0E0") == 1.0 and M.stof("-1.0") == -1.0
    assert M.stof(too_small) == np.float32(np.longdouble(too_small))  # == 0.f
    assert math.isnan(M.stof("NAN")) and math.isinf(M.stof("INF"))
    assert M.stod("1.0") == 1.0 and M.stod("1.0E0") == 1.0 and M.stod("-1.0") == -1.0
    assert M.stod(too_small) == float(np.longdouble(too_small)) and M.stod(too_small) != 0.0  # a denormal double
    assert math.isnan(M.stod("NAN")) and math.isinf(M.stod("INF"))
    for fn in (M.stof, M.stod):
        with pytest.raises(ValueError, match="std::out_of_range"):
            fn("1.23E-1000000000")
        with pytest.raises(ValueError, match="std::invalid_argument"):
            fn("hello")


def test_batch_buffer_decode_pair(small_problem):
    # test_HMM.cpp:44-51
    hmm, _ = _hmm(small_problem)
    assert len(hmm.getBatchBuffer()) == 0
    hmm.decodePair(0, 9)
    assert len(hmm.getBatchBuffer()) == 4
    hmm.decodePair(1, 1)
    assert len(hmm.getBatchBuffer()) == 5


def test_batch_buffer_decode_pairs(small_problem):
    # test_HMM.cpp:53-58
    hmm, _ = _hmm(small_problem)
    assert len(hmm.getBatchBuffer()) == 0
    hmm.decodePairs([0, 1], [9, 1])
    buf = hmm.getBatchBuffer()
    assert len(buf) == 5
    # the buffer holds the pairs' observations (HMM.hpp:215): XOR / AND of the two haplotypes (HMM.cpp:129-157)
    want = hmm.makePairObs(1, 0, 1, 9)
    assert list(buf[0].obsBits) == list(want.obsBits) and list(buf[0].homMinorBits) == list(want.homMinorBits)


def test_batch_buffer_fill_up(small_problem):
    # test_HMM.cpp:69-78: "default batch size is 64 ... buffer should be empty now"
    hmm, _ = _hmm(small_problem)
    for i in range(1, 64 // 4 + 1):
        hmm.decodePair(0, i)
    assert len(hmm.getBatchBuffer()) == 0
    assert hmm.getQueuedPairs() == 64  # the full batch waits in the work list for the next launch
    hmm.decodePair(0, 17)
    assert len(hmm.getBatchBuffer()) == 4


def test_individual_and_package_names(small_problem):
    from asmc import (ASMC, BinaryDataReader, Data, DecodePairsReturnStruct, DecodingMode,  # noqa: F401
                      DecodingModeOverall, DecodingParams, DecodingQuantities, DecodingReturnValues, FastSMC, HMM,
                      IbdPairDataLine, Individual, PairObservations)
    from asmc.pyASMC import HMM as HMM2

    assert HMM is api.HMM and HMM2 is api.HMM and Individual is M.Individual
    ind = Individual(6)  # Individual.cpp:18-22: both genotypes all false
    assert list(ind.genotype1) == [False] * 6 and list(ind.genotype2) == [False] * 6
    ind.setGenotype(1, 2, True)  # hap 1 -> genotype1, anything else -> genotype2 (Individual.cpp:25-32)
    ind.setGenotype(2, 4, True)
    assert list(ind.genotype1) == [0, 0, 1, 0, 0, 0] and list(ind.genotype2) == [0, 0, 0, 0, 1, 0]
    assert len(Individual().genotype1) == 0
    # Data.individuals (Data.hpp:36): the two haplotype rows of every sample
    _, data = _hmm(small_problem)
    inds = data.individuals
    assert len(inds) == 32
    assert list(inds[3].genotype1) == list(data.genotype(6)) and list(inds[3].genotype2) == list(data.genotype(7))


@pytest.mark.gpu
def test_batch_buffer_finish_decoding(small_problem):
    # test_HMM.cpp:60-67
    hmm, _ = _hmm(small_problem, doPosteriorSums=True)
    assert len(hmm.getBatchBuffer()) == 0
    hmm.decodePair(0, 9)
    assert len(hmm.getBatchBuffer()) == 4
    hmm.finishDecoding()
    assert len(hmm.getBatchBuffer()) == 0 and hmm.getQueuedPairs() == 0
    s = hmm.getDecodingReturnValues().sumOverPairs
    np.testing.assert_allclose(s.sum(axis=1), 4.0, rtol=1e-5)  # four posteriors per site, each summing to 1


@pytest.mark.gpu
def test_results_are_current_after_a_full_batch(small_problem):
    """The reference decodes a batch the moment it fills up (addToBatch, HMM.cpp:555-590); here the getter decodes
    what is waiting, so a caller that reads sumOverPairs after 64 queued pairs sees them."""
    hmm, _ = _hmm(small_problem, doPosteriorSums=True)
    for i in range(1, 17):
        hmm.decodePair(0, i)
    s = hmm.getDecodingReturnValues().sumOverPairs
    np.testing.assert_allclose(s.sum(axis=1), 64.0, rtol=1e-5)
    assert hmm.getQueuedPairs() == 0


@pytest.mark.gpu
def test_decode_summarize(small_problem):
    # test_HMM.cpp:35-42 + the definition HMM.cpp:1498-1517 evaluated on the posterior of HMM::decode
    hmm, _ = _hmm(small_problem)
    obs = hmm.makePairObs(1, 0, 2, 0)
    post = np.array(hmm.decode(obs), np.float32)  # [K][S]
    mp, mean = hmm.decodeSummarize(obs)
    assert len(mp) == len(mean) == post.shape[1]
    et = np.array(hmm.getDecodingQuantities().expectedTimes, np.float32)
    want_mean = np.zeros(post.shape[1], np.float32)
    for k in range(post.shape[0]):  # posterior_mean[j] += posterior[i][j] * expectedTimes[i], i ascending
        want_mean = want_mean + post[k] * et[k]
    np.testing.assert_array_equal(np.array(mean, np.float32), want_mean)
    np.testing.assert_array_equal(np.array(mp, np.float32), et[np.argmax(post, axis=0)])  # first maximum
