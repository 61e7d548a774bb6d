// decoding_params.hpp -- option struct of the decode path.  Field names are the reference's
// (DecodingParams.hpp:34-86): they are part of the Python API (pybind.cpp:121-179).  Command-line
// parsing is out of scope; API misuse throws instead of exit(1).
#pragma once

#include <string>

namespace fsmc_host
{

enum class DecodingMode { sequenceFolded, arrayFolded, sequence, array };
enum class DecodingModeOverall { sequence, array };

class DecodingParams
{
public:
  std::string inFileRoot;
  std::string decodingQuantFile;
  std::string outFileRoot;
  int jobs = 1;
  int jobInd = 1;
  std::string decodingModeString = "array";
  DecodingModeOverall decodingModeOverall = DecodingModeOverall::array;
  DecodingMode decodingMode = DecodingMode::arrayFolded;
  bool decodingSequence = false;
  bool foldData = false;
  bool usingCSFS = false;
  bool compress = false;
  bool useAncestral = false;
  float skipCSFSdistance = 0.f;
  bool noBatches = false;

  int batchSize = 64;
  int recallThreshold = 3;
  float skip = 0.f;
  int gap = 1;
  int max_seeds = 0;
  float min_maf = 0;
  float min_m = 1;
  bool hashing = false;
  bool FastSMC = false;
  bool BIN_OUT = false;
  bool useKnownSeed = false;
  bool outputIbdSegmentLength = false;
  int hashingWordSize = 64;
  int constReadAhead = 10;
  bool haploid = true;
  int time = 100;

  bool noConditionalAgeEstimates = false;
  bool doPosteriorSums = false;
  bool doPerPairPosteriorMean = false;
  bool doPerPairMAP = false;
  std::string expectedCoalTimesFile;
  bool withinOnly = false;
  bool doMajorMinorPosteriorSums = false;

  // MI355X additions (not in the reference): which device this process drives.
  int gpuDevice = 0;

  DecodingParams();
  // ASMC-style constructor (DecodingParams.cpp:39-54)
  explicit DecodingParams(std::string _inFileRoot, std::string _decodingQuantFile = "", std::string _outFileRoot = "",
                          int _jobs = 1, int _jobInd = 1, std::string _decodingModeString = "array",
                          bool _decodingSequence = false, bool _usingCSFS = true, bool _compress = false,
                          bool _useAncestral = false, float _skipCSFSdistance = 0.f, bool _noBatches = false,
                          bool _doPosteriorSums = false, bool _doPerPairPosteriorMean = false,
                          std::string _expectedCoalTimesFile = "", bool _withinOnly = false,
                          bool _doMajorMinorPosteriorSums = false, bool _doPerPairMAP = false);
  // FastSMC defaults (DecodingParams.cpp:56-73)
  DecodingParams(std::string _inFileRoot, std::string _decodingQuantFile, std::string _outFileRoot, bool _fastSMC);

  bool processOptions();       // DecodingParams.cpp:466-558
  bool validateParamsFastSMC(); // DecodingParams.cpp:278-464 (without the option dump)
};

} // namespace fsmc_host
