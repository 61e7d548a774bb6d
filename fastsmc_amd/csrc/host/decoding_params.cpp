#include "decoding_params.hpp"

#include <cmath>
#include <limits>
#include <stdexcept>

#include "util.hpp"

namespace fsmc_host
{

DecodingParams::DecodingParams() : usingCSFS(true) {}

DecodingParams::DecodingParams(std::string _inFileRoot, std::string _decodingQuantFile, std::string _outFileRoot,
                               int _jobs, int _jobInd, std::string _decodingModeString, bool _decodingSequence,
                               bool _usingCSFS, bool _compress, bool _useAncestral, float _skipCSFSdistance,
                               bool _noBatches, bool _doPosteriorSums, bool _doPerPairPosteriorMean,
                               std::string _expectedCoalTimesFile, bool _withinOnly, bool _doMajorMinorPosteriorSums,
                               bool _doPerPairMAP)
    : inFileRoot(std::move(_inFileRoot)), decodingQuantFile(std::move(_decodingQuantFile)),
      outFileRoot(std::move(_outFileRoot)), jobs(_jobs), jobInd(_jobInd),
      decodingModeString(std::move(_decodingModeString)), decodingSequence(_decodingSequence), usingCSFS(_usingCSFS),
      compress(_compress), useAncestral(_useAncestral), skipCSFSdistance(_skipCSFSdistance), noBatches(_noBatches),
      doPosteriorSums(_doPosteriorSums), doPerPairPosteriorMean(_doPerPairPosteriorMean), doPerPairMAP(_doPerPairMAP),
      expectedCoalTimesFile(std::move(_expectedCoalTimesFile)), withinOnly(_withinOnly),
      doMajorMinorPosteriorSums(_doMajorMinorPosteriorSums)
{
  if (!processOptions()) {
    throw std::runtime_error("invalid decoding parameters");
  }
}

DecodingParams::DecodingParams(std::string _inFileRoot, std::string _decodingQuantFile, std::string _outFileRoot,
                               bool _fastSMC)
    : inFileRoot(std::move(_inFileRoot)), decodingQuantFile(std::move(_decodingQuantFile)),
      outFileRoot(std::move(_outFileRoot)), foldData(true), usingCSFS(true), batchSize(32), recallThreshold(3),
      min_m(1.5f), hashing(true), FastSMC(_fastSMC), BIN_OUT(false), outputIbdSegmentLength(true), time(50),
      noConditionalAgeEstimates(true), doPerPairPosteriorMean(true), doPerPairMAP(true)
{
  if (!FastSMC) {
    throw std::runtime_error("This DecodingParams constructor sets FastSMC defaults and is only intended for use "
                             "with FastSMC. Set the fastSMC parameter to true, or use a different constructor.");
  }
  validateParamsFastSMC();
}

namespace
{
// shared tail of both validators: mode string -> enums / folding (DecodingParams.cpp:330-357, 499-527)
void resolveMode(DecodingParams& p)
{
  p.decodingModeString = toLower(p.decodingModeString);
  if (p.decodingModeString == "sequence") {
    p.decodingModeOverall = DecodingModeOverall::sequence;
    p.decodingSequence = true;
    p.decodingMode = p.useAncestral ? DecodingMode::sequence : DecodingMode::sequenceFolded;
  } else if (p.decodingModeString == "array") {
    p.decodingModeOverall = DecodingModeOverall::array;
    p.decodingSequence = false;
    p.decodingMode = p.useAncestral ? DecodingMode::array : DecodingMode::arrayFolded;
  } else {
    throw std::runtime_error("Decoding mode should be one of {sequence, array}, got " + p.decodingModeString);
  }
  p.foldData = !p.useAncestral;
}

void resolveCompress(DecodingParams& p)
{
  if (p.compress) {
    if (p.useAncestral) {
      throw std::runtime_error("compress & useAncestral cannot be used together. A compressed emission cannot use "
                               "ancestral allele information.");
    }
    if (!std::isnan(p.skipCSFSdistance) && p.skipCSFSdistance != 0.f &&
        p.skipCSFSdistance != std::numeric_limits<float>::infinity()) {
      throw std::runtime_error("compress & skipCSFSdistance cannot be used together. compress is a shorthand for "
                               "skipCSFSdistance Infinity.");
    }
    p.skipCSFSdistance = std::numeric_limits<float>::infinity();
  } else if (std::isnan(p.skipCSFSdistance)) {
    p.skipCSFSdistance = 0.f;
  }
  if (p.skipCSFSdistance != std::numeric_limits<float>::infinity()) {
    p.usingCSFS = true;
  }
}
} // namespace

bool DecodingParams::processOptions()
{
  resolveCompress(*this);
  if (!expectedCoalTimesFile.empty()) {
    doPerPairPosteriorMean = true;
  }
  resolveMode(*this);
  if (decodingQuantFile.empty()) {
    decodingQuantFile = inFileRoot + ".decodingQuantities.bin";
  }
  if ((jobs == 0) != (jobInd == 0)) {
    return false;
  }
  if (jobs == 0) {
    jobs = 1;
    jobInd = 1;
  }
  if (jobInd <= 0 || jobInd > jobs) {
    return false;
  }
  if (outFileRoot.empty()) {
    outFileRoot = inFileRoot + "." + std::to_string(jobInd) + "-" + std::to_string(jobs);
  }
  return true;
}

bool DecodingParams::validateParamsFastSMC()
{
  if (!FastSMC) {
    throw std::runtime_error("Attempting to validate FastSMC parameters but FastSMC flag is false.");
  }
  if (hashing) {
    if (withinOnly) {
      throw std::runtime_error("hashing & withinOnly cannot be used together.");
    }
    if (time <= 0) {
      throw std::runtime_error("time must be a positive integer.");
    }
  }
  if (batchSize == 0 || batchSize % 8 != 0) {
    throw std::runtime_error("batchSize must be strictly positive and a multiple of 8.");
  }
  resolveCompress(*this);
  resolveMode(*this);
  if (decodingQuantFile.empty()) {
    decodingQuantFile = inFileRoot + ".decodingQuantities.bin";
  }
  if ((jobs == 0) != (jobInd == 0)) {
    throw std::runtime_error("jobs and jobInd must either both be set or both be unset");
  }
  if (jobs == 0) {
    jobs = 1;
    jobInd = 1;
  }
  if (jobInd <= 0 || jobInd > jobs || jobs <= 0) {
    throw std::runtime_error("jobInd must be between 1 and jobs inclusive");
  }
  // jobs must be a square number: 1, 4, 9, ... (DecodingParams.cpp:376-395)
  bool validJob = false;
  for (int x = 1, u = 1, i = 0; i < 200 && u <= jobs; ++i) {
    if (u == jobs) {
      validJob = true;
      break;
    }
    x += 2;
    u += x;
  }
  if (!validJob) {
    throw std::runtime_error("jobs value is incorrect: it must be a perfect square");
  }
  if (recallThreshold < 0 || recallThreshold > 3) {
    throw std::runtime_error("recall must be between 0 and 3.");
  }
  if (outFileRoot.empty()) {
    outFileRoot = inFileRoot + "." + std::to_string(jobInd) + "-" + std::to_string(jobs);
  }
  return true;
}

} // namespace fsmc_host
