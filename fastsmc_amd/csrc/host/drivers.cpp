#include "drivers.hpp"

#include "hashing.hpp"

#include <stdexcept>

namespace fsmc_host
{

namespace
{
// (in the member-initialiser list, before Data reads the files: the device comes up while the host parses)
DecodingParams withDeviceStarting(DecodingParams p)
{
  warmUpDevice(p.gpuDevice);
  return p;
}
} // namespace

ASMC::ASMC(DecodingParams params) : mParams(withDeviceStarting(std::move(params))), mHmm(Data(mParams), mParams) {}

ASMC::ASMC(const std::string& inFileRoot, const std::string& decodingQuantFile, const std::string& outFileRoot)
    : mParams(inFileRoot, decodingQuantFile, outFileRoot.empty() ? inFileRoot : outFileRoot, 1, 1, "array", false,
              true, false, false, 0.f, false, true, false, "", false, true, true),
      mHmm(Data(mParams), mParams)
{
}

DecodingReturnValues ASMC::decodeAllInJob()
{
  mHmm.decodeAll(mParams.jobs, mParams.jobInd);
  return mHmm.getDecodingReturnValues();
}

void ASMC::decodePairs(const std::vector<unsigned long>& hapIndicesA, const std::vector<unsigned long>& hapIndicesB,
                       bool perPairPosteriors, bool sumOfPosteriors, bool perPairPosteriorMeans, bool perPairMAPs)
{
  if (hapIndicesA.empty() || hapIndicesA.size() != hapIndicesB.size()) {
    throw std::runtime_error("Vector of A indices (" + std::to_string(hapIndicesA.size()) +
                             ") must be the same size as vector of B indices (" +
                             std::to_string(hapIndicesB.size()) + ").\n");
  }
  mHmm.getDecodePairsReturnStruct().initialise(hapIndicesA, hapIndicesB, mHmm.getData().sites,
                                               mHmm.getDecodingQuantities().states, perPairPosteriors, sumOfPosteriors,
                                               perPairPosteriorMeans, perPairMAPs);
  mHmm.setStorePerPairPosteriorMean(perPairPosteriorMeans);
  mHmm.setStorePerPairMap(perPairMAPs);
  mHmm.setStorePerPairPosterior(perPairPosteriors);
  mHmm.setStoreSumOfPosterior(sumOfPosteriors);
  mHmm.decodeHapPairs(hapIndicesA, hapIndicesB);
  mHmm.finishDecoding();
  mHmm.getDecodePairsReturnStruct().finaliseCalculations();
}

void ASMC::decodePairs(const std::vector<std::string>& hapIdsA, const std::vector<std::string>& hapIdsB,
                       bool perPairPosteriors, bool sumOfPosteriors, bool perPairPosteriorMeans, bool perPairMAPs)
{
  if (hapIdsA.size() != hapIdsB.size()) {
    throw std::runtime_error("Vector of A IDs (" + std::to_string(hapIdsA.size()) +
                             ") must be the same size as vector of B IDs (" + std::to_string(hapIdsB.size()) + ").\n");
  }
  std::vector<unsigned long> a(hapIdsA.size()), b(hapIdsB.size());
  const auto& ids = mHmm.getData().IIDList;
  for (size_t i = 0; i < hapIdsA.size(); ++i) {
    const auto [strA, hapA] = combinedIdToIndPlusHap(hapIdsA[i]);
    const auto [strB, hapB] = combinedIdToIndPlusHap(hapIdsB[i]);
    a[i] = dipToHapId(getIndIdxFromIdString(ids, strA), hapA);
    b[i] = dipToHapId(getIndIdxFromIdString(ids, strB), hapB);
  }
  decodePairs(a, b, perPairPosteriors, sumOfPosteriors, perPairPosteriorMeans, perPairMAPs);
}

FastSMC::FastSMC(DecodingParams params) : mParams(withDeviceStarting(std::move(params))), mHmm(Data(mParams), mParams) {}

FastSMC::FastSMC(const std::string& inFileRoot, const std::string& outFileRoot)
    : mParams(withDeviceStarting(DecodingParams(inFileRoot, inFileRoot + ".decodingQuantities.gz", outFileRoot, true))),
      mHmm(Data(mParams), mParams)
{
}

void FastSMC::run()
{
  mHmm.decodeAll(mParams.jobs, mParams.jobInd);
  if (!mParams.hashing) {
    mHmm.closeIBDFile();
    return;
  }
  // identification step: candidate (pair, window)s -> HMM::decodeFromHashing (FastSMC.cpp:118-235)
  runHashing(mHmm.getData(), mParams, mHmm);
  mHmm.finishFromHashing();
}

} // namespace fsmc_host
