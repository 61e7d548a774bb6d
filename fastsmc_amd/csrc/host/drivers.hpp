// drivers.hpp -- the two driver classes of the reference's API: ASMC (pair-list decode, ASMC.hpp/.cpp)
// and FastSMC (IBD detection, FastSMC.hpp/.cpp), as thin owners of Data + HMM.
#pragma once

#include <string>
#include <vector>

#include "hmm.hpp"

namespace fsmc_host
{

class ASMC
{
public:
  explicit ASMC(DecodingParams params);
  // ASMC.cpp:28-49: array mode, posterior sums + per-pair mean + MAP enabled
  ASMC(const std::string& inFileRoot, const std::string& decodingQuantFile, const std::string& outFileRoot = "");

  DecodingReturnValues decodeAllInJob(); // ASMC.cpp:51-78
  void decodePairs(const std::vector<unsigned long>& hapIndicesA, const std::vector<unsigned long>& hapIndicesB,
                   bool perPairPosteriors = false, bool sumOfPosteriors = false, bool perPairPosteriorMeans = false,
                   bool perPairMAPs = false); // ASMC.cpp:80-100
  void decodePairs(const std::vector<std::string>& hapIdsA, const std::vector<std::string>& hapIdsB,
                   bool perPairPosteriors = false, bool sumOfPosteriors = false, bool perPairPosteriorMeans = false,
                   bool perPairMAPs = false); // ASMC.cpp:102-128
  DecodePairsReturnStruct getCopyOfResults() { return mHmm.getDecodePairsReturnStruct(); }
  const DecodePairsReturnStruct& getRefOfResults() { return mHmm.getDecodePairsReturnStruct(); }
  HMM& hmm() { return mHmm; }

private:
  DecodingParams mParams;
  HMM mHmm;
};

class FastSMC
{
public:
  explicit FastSMC(DecodingParams params);
  // FastSMC.cpp:34-39: decoding quantities at <in>.decodingQuantities.gz, FastSMC defaults
  FastSMC(const std::string& inFileRoot, const std::string& outFileRoot);
  void run(); // FastSMC.cpp:41-238
  HMM& hmm() { return mHmm; }
  std::string outputFileName() const { return mHmm.ibdFileName(mParams.jobs, mParams.jobInd); }

private:
  DecodingParams mParams;
  HMM mHmm;
};

} // namespace fsmc_host
