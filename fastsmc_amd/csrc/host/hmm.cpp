// hmm.cpp -- host orchestration of the pairwise decode (counterpart of the reference's HMM.cpp).
// Everything numerical on the hot path happens in libfastsmc_hip.so; this file prepares its constant
// inputs exactly like HMM::HMM does, turns the reference's pair/batch bookkeeping into a GPU work
// list, and writes results in the reference's formats and order.
#include "hmm.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstdio>
#include <chrono>
#include <future>
#include <mutex>
#include <cmath>
#include <iomanip>
#include <limits>
#include <map>
#include <sstream>
#include <stdexcept>
#include <thread>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include "util.hpp"

namespace fsmc_host
{

// ------------------------------------------------------------------ HmmUtils counterparts

float roundMorgans(const float value, const int precision, const float min)
{
  if (value <= min) {
    return min;
  }
  const float correction = 10.f - static_cast<float>(precision);
  const float L10 = std::max<float>(0.f, floorf(log10f(value)) + correction);
  const float factor = powf(10.f, 10.f - L10);
  return roundf(value * factor) / factor;
}

int roundPhysical(const int value, const int precision)
{
  if (value <= 1) {
    return 1;
  }
  const int L10 = std::max<int>(0, static_cast<int>(floor(log10(value))) - precision);
  const int factor = static_cast<int>(pow(10, L10));
  return static_cast<int>(round(value / static_cast<double>(factor))) * factor;
}

unsigned getFromPosition(const std::vector<float>& gen, unsigned from, const float cmDist)
{
  float cum = 0.f;
  while (cum < cmDist && from > 0u) {
    from--;
    cum += (gen[from + 1u] - gen[from]) * 100.f;
  }
  return from;
}

unsigned getToPosition(const std::vector<float>& gen, unsigned to, const float cmDist)
{
  float cum = 0.f;
  while (cum < cmDist && to + 1u < gen.size()) {
    to++;
    cum += (gen[to] - gen[to - 1u]) * 100.f;
  }
  return std::min<unsigned>(to + 1u, static_cast<unsigned>(gen.size()));
}

std::pair<unsigned long, unsigned long> hapToDipId(unsigned long hapId)
{
  return {hapId / 2ul, 1ul + (hapId % 2ul)};
}

unsigned long dipToHapId(unsigned long ind, unsigned long hap)
{
  return 2ul * ind + hap - 1ul;
}

std::string indPlusHapToCombinedId(const std::string& indId, unsigned long hap)
{
  if (indId.empty() || !(hap == 1ul || hap == 2ul)) {
    throw std::runtime_error("Expected an individual ID and either 1 or 2, but got " + indId + " and " +
                             std::to_string(hap) + "\n");
  }
  return indId + "#" + std::to_string(hap);
}

std::pair<std::string, unsigned long> combinedIdToIndPlusHap(const std::string& id)
{
  const size_t n = id.length();
  if (n < 3 || !(id.compare(n - 2, 2, "#1") == 0 || id.compare(n - 2, 2, "#2") == 0)) {
    throw std::runtime_error("Expected combined ID in form <id>#1 OR <id>#2, but got " + id + "\n");
  }
  return {id.substr(0, n - 2), id.back() == '1' ? 1ul : 2ul};
}

unsigned long getIndIdxFromIdString(const std::vector<std::string>& ids, const std::string& id)
{
  auto it = std::find(ids.begin(), ids.end(), id);
  if (it == ids.end()) {
    throw std::runtime_error("The ID string " + id + " is not in the list of IDs\n");
  }
  return static_cast<unsigned long>(std::distance(ids.begin(), it));
}

// ------------------------------------------------------------------ return structs

void DecodePairsReturnStruct::initialise(const std::vector<unsigned long>& hapsA, const std::vector<unsigned long>&,
                                         long sites, long states, bool fullPosteriors, bool sumOfPost,
                                         bool perPairMeans, bool perPairMaps)
{
  numWritten = 0;
  numPairs = static_cast<long>(hapsA.size());
  numSites = sites;
  numStates = states;
  storeFullPosteriors = fullPosteriors;
  storeSumOfPosteriors = sumOfPost;
  storePerPairPosteriorMeans = perPairMeans;
  storePerPairMAPs = perPairMaps;
  perPairIndices.assign(static_cast<size_t>(numPairs), {});
  perPairPosteriors.clear();
  sumOfPosteriors.clear();
  perPairPosteriorMeans.clear();
  perPairMAPs.clear();
  minPosteriorMeans.clear();
  argminPosteriorMeans.clear();
  minMAPs.clear();
  argminMAPs.clear();
  if (fullPosteriors) {
    perPairPosteriors.assign(static_cast<size_t>(numPairs), std::vector<float>(static_cast<size_t>(states * sites)));
  }
  if (sumOfPost) {
    sumOfPosteriors.assign(static_cast<size_t>(states * sites), 0.f);
  }
  if (perPairMeans) {
    perPairPosteriorMeans.assign(static_cast<size_t>(numPairs * sites), 0.f);
    minPosteriorMeans.assign(static_cast<size_t>(sites), 0.f);
    argminPosteriorMeans.assign(static_cast<size_t>(sites), 0);
  }
  if (perPairMaps) {
    perPairMAPs.assign(static_cast<size_t>(numPairs * sites), 0);
    minMAPs.assign(static_cast<size_t>(sites), 0);
    argminMAPs.assign(static_cast<size_t>(sites), 0);
  }
}

void DecodePairsReturnStruct::finaliseCalculations()
{
  // column-wise min / first argmin (DecodePairsReturnStruct.hpp:105-118)
  if (!perPairPosteriorMeans.empty()) {
    for (long s = 0; s < numSites; ++s) {
      long arg = 0;
      float best = perPairPosteriorMeans[static_cast<size_t>(s)];
      for (long p = 1; p < numPairs; ++p) {
        const float v = perPairPosteriorMeans[static_cast<size_t>(p * numSites + s)];
        if (v < best) {
          best = v;
          arg = p;
        }
      }
      minPosteriorMeans[static_cast<size_t>(s)] = best;
      argminPosteriorMeans[static_cast<size_t>(s)] = static_cast<int>(arg);
    }
  }
  if (!perPairMAPs.empty()) {
    for (long s = 0; s < numSites; ++s) {
      long arg = 0;
      int best = perPairMAPs[static_cast<size_t>(s)];
      for (long p = 1; p < numPairs; ++p) {
        const int v = perPairMAPs[static_cast<size_t>(p * numSites + s)];
        if (v < best) {
          best = v;
          arg = p;
        }
      }
      minMAPs[static_cast<size_t>(s)] = best;
      argminMAPs[static_cast<size_t>(s)] = static_cast<int>(arg);
    }
  }
}

// ------------------------------------------------------------------ construction

namespace
{
void check(fsmc_ctx* ctx, int rc, const char* what)
{
  if (rc != FSMC_OK) {
    throw std::runtime_error(std::string(what) + " failed (" + std::to_string(rc) + "): " + fsmc_last_error(ctx));
  }
}
} // namespace

HMM::HMM(Data data, const DecodingParams& params, int scalingSkip)
    : mData(std::move(data)), mDq(params.decodingQuantFile), mParams(params)
{
  init(scalingSkip);
}

HMM::HMM(Data data, DecodingQuantities dq, const DecodingParams& params, int scalingSkip)
    : mData(std::move(data)), mDq(std::move(dq)), mParams(params)
{
  init(scalingSkip);
}

void HMM::init(int scalingSkip)
{
  if (mParams.hashing && !mParams.FastSMC) {
    throw std::runtime_error("Identification only is not yet supported: cannot have hashing==true and "
                             "FastSMC==false.");
  }
  if (scalingSkip != 1) {
    // every caller in the reference uses 1 (HMM.hpp:172, pybind.cpp:219); the GPU path rescales every site
    throw std::runtime_error("scalingSkip != 1 is not supported");
  }
  if (mParams.noBatches) {
    // the reference's noBatches runs its scalar debugging path; here every decode is batched on the GPU
    mParams.noBatches = false;
  }
  mBatchSize = mParams.batchSize;
  if (mBatchSize <= 0) {
    throw std::runtime_error("batchSize must be positive");
  }
  mFromBatch.assign(static_cast<size_t>(mBatchSize), 0u);
  mToBatch.assign(static_cast<size_t>(mBatchSize), static_cast<unsigned>(mData.sites));
  prepareModel();
  mReturn.sites = mData.sites;
  mReturn.states = mDq.states;
  mReturn.siteWasFlippedDuringFolding = mData.siteWasFlippedDuringFolding;
  resetDecoding();
}

HMM::~HMM()
{
  closePerPairFiles();
  if (mIbdFile) {
    gzclose(mIbdFile);
  }
  if (mModel) {
    fsmc_model_destroy(mModel);
  }
  if (mCtx) {
    fsmc_ctx_destroy(mCtx);
  }
}

void HMM::prepareEmissions()
{
  // HMM.cpp:159-256
  const bool seq = mParams.decodingSequence;
  const int S = mData.sites;
  const int K = static_cast<int>(mDq.states);
  const auto undist = mData.calculateUndistinguishedCounts(mDq.CSFSSamples);
  mUseCSFS.assign(static_cast<size_t>(S), false);
  if (mParams.skipCSFSdistance < std::numeric_limits<float>::infinity()) {
    mUseCSFS[0] = true;
    float lastGenCSFSwasUsed = 0.f;
    for (int pos = 1; pos < S; pos++) {
      if (mData.geneticPositions[pos] - lastGenCSFSwasUsed >= mParams.skipCSFSdistance) {
        mUseCSFS[pos] = true;
        lastGenCSFSwasUsed = mData.geneticPositions[pos];
      }
    }
  }
  mPrep.e1.assign(static_cast<size_t>(S) * K, 0.f);
  mPrep.e0m1.assign(static_cast<size_t>(S) * K, 0.f);
  mPrep.e2m0.assign(static_cast<size_t>(S) * K, 0.f);
  auto csfsRow = [&](const std::vector<std::vector<std::vector<float>>>& map, int u, int d) -> const float* {
    if (u < 0 || static_cast<size_t>(u) >= map.size() || map[u].size() <= static_cast<size_t>(d) ||
        map[u][d].size() != static_cast<size_t>(K)) {
      throw std::runtime_error("decoding quantities lack the CSFS row for undistinguished count " +
                               std::to_string(u));
    }
    return map[u][d].data();
  };
  for (int pos = 0; pos < S; pos++) {
    float* e1 = &mPrep.e1[static_cast<size_t>(pos) * K];
    float* e0m1 = &mPrep.e0m1[static_cast<size_t>(pos) * K];
    float* e2m0 = &mPrep.e2m0[static_cast<size_t>(pos) * K];
    if (mUseCSFS[pos]) {
      const int u0 = undist[pos][0], u1 = undist[pos][1], u2 = undist[pos][2];
      if (mParams.foldData) {
        const auto& map = seq ? mDq.foldedCSFSmap : mDq.foldedAscertainedCSFSmap;
        const float* r1 = u1 >= 0 ? csfsRow(map, u1, 1) : nullptr;
        const float* r0 = csfsRow(map, u0, 0);
        const float* r2 = u2 >= 0 ? csfsRow(map, u2, 0) : nullptr;
        for (int k = 0; k < K; k++) {
          e1[k] = r1 ? r1[k] : 0.f;
          e0m1[k] = r0[k] - e1[k];
          e2m0[k] = r2 ? (r2[k] - r0[k]) : (0 - r0[k]);
        }
      } else {
        const auto& map = seq ? mDq.CSFSmap : mDq.ascertainedCSFSmap;
        const float* r1 = u1 >= 0 ? csfsRow(map, u1, 1) : nullptr;
        const float* r0 = u0 >= 0 ? csfsRow(map, u0, 0) : nullptr;
        const float* r2 = nullptr;
        if (u2 >= 0) {
          // monomorphic derived folds to CSFS[0][0] (HMM.cpp:226-232)
          r2 = (u2 == mDq.CSFSSamples - 2) ? csfsRow(map, 0, 0) : csfsRow(map, u2, 2);
        }
        for (int k = 0; k < K; k++) {
          e1[k] = r1 ? r1[k] : 0.f;
          const float e0 = r0 ? r0[k] : 0.f;
          e0m1[k] = e0 - e1[k];
          e2m0[k] = r2 ? (r2[k] - e0) : (0 - e0);
        }
      }
    } else {
      const auto& table = seq ? mDq.classicEmissionTable : mDq.compressedEmissionTable;
      if (table.size() != 2) {
        throw std::runtime_error(seq ? "decoding quantities lack the ClassicEmission table"
                                     : "decoding quantities lack the CompressedAscertainedEmission table");
      }
      const float* c0 = table[0].data();
      const float* c1 = table[1].data();
      for (int k = 0; k < K; k++) {
        e1[k] = c1[k];
        e0m1[k] = c0[k] - c1[k];
        e2m0[k] = 0.f;
      }
    }
  }
}

void HMM::prepareModel()
{
  const int S = mData.sites;
  const int K = static_cast<int>(mDq.states);
  if (S < 1) {
    throw std::runtime_error("no sites to decode");
  }
  if (static_cast<int>(mDq.initialStateProb.size()) != K || static_cast<int>(mDq.expectedTimes.size()) != K ||
      static_cast<int>(mDq.discretization.size()) != K + 1 || static_cast<int>(mDq.columnRatios.size()) != K) {
    throw std::runtime_error("decoding quantities are incomplete (initialStateProb / expectedTimes / "
                             "discretization / columnRatios)");
  }
  mPrep.K = K;
  mPrep.S = S;
  mPrep.pi = mDq.initialStateProb;
  mPrep.colRatios = mDq.columnRatios;
  mPrep.expTimes = mDq.expectedTimes;
  prepareEmissions();

  // Transition-table row per site step: key = roundMorgans(gen[p] - gen[p-1]) in fp32, exact-match lookup
  // (HMM.cpp:755, 795-797, 909, 951-954).  Only the rows this data set touches are kept.
  const int precision = 2;
  const float minGenetic = 1e-10f;
  mPrep.stepRow.assign(static_cast<size_t>(S), 0);
  std::map<uint32_t, int> rowOfKey;
  std::vector<float> keys;
  auto rowFor = [&](float key) {
    auto it = rowOfKey.find(floatBits(key));
    if (it == rowOfKey.end()) {
      it = rowOfKey.emplace(floatBits(key), static_cast<int>(keys.size())).first;
      keys.push_back(key);
    }
    return it->second;
  };
  mPrep.sequence = mParams.decodingSequence;
  if (mPrep.sequence) {
    mPrep.gapRowF.assign(static_cast<size_t>(S), 0);
    mPrep.siteRowF.assign(static_cast<size_t>(S), 0);
    mPrep.gapRowB.assign(static_cast<size_t>(S), 0);
    mPrep.siteRowB.assign(static_cast<size_t>(S), 0);
    mPrep.hom.assign(static_cast<size_t>(S) * K, 0.f);
    if (static_cast<int>(mData.recRateAtMarker.size()) != S || static_cast<int>(mData.physicalPositions.size()) != S) {
      throw std::runtime_error("sequence mode needs recombination rates and physical positions for every site");
    }
  }
  for (int p = 1; p < S; ++p) {
    const float key = roundMorgans(mData.geneticPositions[p] - mData.geneticPositions[p - 1], precision, minGenetic);
    mPrep.stepRow[p] = rowFor(key);
    if (mPrep.sequence) {
      // forward into site p (HMM.cpp:755-770): the rate of site p; backward out of site p, i.e. the step that
      // computes beta of p-1 (HMM.cpp:909-925): the rate of site p-1
      const float rateHere = roundMorgans(mData.recRateAtMarker[p], precision, minGenetic);
      const float ratePrev = roundMorgans(mData.recRateAtMarker[p - 1], precision, minGenetic);
      mPrep.gapRowF[p] = rowFor(roundMorgans(key - rateHere, precision, minGenetic));
      mPrep.siteRowF[p] = rowFor(rateHere);
      mPrep.gapRowB[p] = rowFor(roundMorgans(key - ratePrev, precision, minGenetic));
      mPrep.siteRowB[p] = rowFor(ratePrev);
      const int physKey = roundPhysical(mData.physicalPositions[p] - mData.physicalPositions[p - 1] - 1, precision);
      const auto it = mDq.homozygousEmissionMap.find(physKey);
      if (it == mDq.homozygousEmissionMap.end() || static_cast<int>(it->second.size()) != K) {
        // the reference throws std::out_of_range from unordered_map::at
        throw std::out_of_range("no HomozygousEmissions entry for physical distance " + std::to_string(physKey));
      }
      std::copy(it->second.begin(), it->second.end(), mPrep.hom.begin() + static_cast<size_t>(p) * K);
    }
  }
  if (keys.empty()) {
    keys.push_back(minGenetic);
  }
  mPrep.nRows = static_cast<int>(keys.size());
  auto gather = [&](const KeyedTable& t, const char* name) {
    std::vector<float> out(keys.size() * static_cast<size_t>(K));
    for (size_t r = 0; r < keys.size(); ++r) {
      const int src = t.find(keys[r]);
      if (src < 0) { // the reference throws std::out_of_range from unordered_map::at
        std::ostringstream os;
        os << std::setprecision(9) << "no " << name << " entry for genetic distance " << keys[r];
        throw std::out_of_range(os.str());
      }
      std::copy(t.row(src, K), t.row(src, K) + K, out.begin() + r * K);
    }
    return out;
  };
  mPrep.D = gather(mDq.Dvectors, "Dvectors");
  mPrep.B = gather(mDq.Bvectors, "Bvectors");
  mPrep.U = gather(mDq.Uvectors, "Uvectors");
  mPrep.RR = gather(mDq.rowRatioVectors, "RowRatios");

  // HMM::getStateThreshold (HMM.cpp:504-513) and the probability threshold (HMM.cpp:96-99)
  unsigned st = 0;
  while (st < mDq.states && mDq.discretization[st] < static_cast<float>(mParams.time)) {
    st++;
  }
  mPrep.stateThreshold = st;
  float pthr = 0.f;
  for (unsigned i = 0; i < st; i++) {
    pthr += mDq.initialStateProb.at(i);
  }
  mPrep.probabilityThreshold = pthr;
  mPrep.ageThreshold = mParams.noConditionalAgeEstimates ? mDq.states : st;
  // expected coalescence times of the per-pair posterior means (HMM.cpp:1736-1748): the second column of the intervals
  // file when one is given (ASMC mode), else the decoding quantities' own
  mExpectedCoalTimes = mDq.expectedTimes;
  if (!mParams.FastSMC && !mParams.expectedCoalTimesFile.empty() && isRegularFile(mParams.expectedCoalTimesFile)) {
    mExpectedCoalTimes = readExpectedTimesFromIntervalsFile(mParams.expectedCoalTimesFile);
    if (mExpectedCoalTimes.size() != mDq.states) {
      throw std::runtime_error(mParams.expectedCoalTimesFile + " has " + std::to_string(mExpectedCoalTimes.size()) +
                               " intervals, the decoding quantities have " + std::to_string(mDq.states) + " states");
    }
  }
}

// HMM.cpp:43-61: "intervalStart expectedCoalescentTime intervalEnd" on every line; the second column
std::vector<float> readExpectedTimesFromIntervalsFile(const std::string& fileName)
{
  LineReader in(fileName);
  std::vector<float> out;
  std::string line;
  while (in.getline(line)) {
    const std::vector<std::string> tok = splitWhitespace(line);
    if (tok.size() != 3) {
      throw std::runtime_error(fileName + " should have \"intervalStart\texpectedCoalescentTime\tintervalEnd\" at each line.");
    }
    out.push_back(refStof(tok[1]));
  }
  return out;
}

bool isRegularFile(const std::string& path)
{
  struct stat st{};
  return ::stat(path.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}

namespace
{
// gzopen(name, mode) that also hands out the descriptor
gzFile gzOpenWithFd(const std::string& name, const char* mode, int& fd)
{
  fd = ::open(name.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
  if (fd < 0) {
    return nullptr;
  }
  gzFile f = gzdopen(fd, mode);
  if (!f) {
    ::close(fd);
    fd = -1;
  }
  return f;
}

// one complete gzip member holding `n` bytes, at zlib's default level (what gzopen(name, "w") writes with)
std::string gzipMember(const char* data, size_t n)
{
  z_stream z{};
  if (deflateInit2(&z, Z_DEFAULT_COMPRESSION, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) {
    throw std::runtime_error("deflateInit2 failed");
  }
  std::string out(deflateBound(&z, static_cast<uLong>(n)) + 32, '\0');
  z.next_in = reinterpret_cast<Bytef*>(const_cast<char*>(data));
  z.avail_in = static_cast<uInt>(n);
  z.next_out = reinterpret_cast<Bytef*>(&out[0]);
  z.avail_out = static_cast<uInt>(out.size());
  const int rc = deflate(&z, Z_FINISH);
  const size_t have = out.size() - z.avail_out;
  deflateEnd(&z);
  if (rc != Z_STREAM_END) {
    throw std::runtime_error("deflate failed");
  }
  out.resize(have);
  return out;
}

// threads that format / compress a flush's output (FSMC_HOST_OUTPUT_THREADS=1 in the environment: the one-thread path,
// for A/B timings)
size_t outputThreads()
{
  static const size_t n = [] {
    if (const char* v = std::getenv("FSMC_HOST_OUTPUT_THREADS")) {
      const int t = std::atoi(v);
      if (t >= 1) {
        return static_cast<size_t>(t);
      }
    }
    return static_cast<size_t>(std::min(16u, std::max(1u, std::thread::hardware_concurrency())));
  }();
  return n;
}

void writeAll(int fd, const std::string& bytes)
{
  size_t off = 0;
  while (off < bytes.size()) {
    const ssize_t w = ::write(fd, bytes.data() + off, bytes.size() - off);
    if (w < 0) {
      throw std::runtime_error("cannot write the output file");
    }
    off += static_cast<size_t>(w);
  }
}

// A flush's text into the IBD file.  Small: gzwrite.  Large: the text is cut at line ends into pieces that several
// threads compress into gzip members of their own, which go to the file behind the member gzwrite had under way (a gzip
// file is a sequence of members; zlib, zcat, Python's gzip and java.util.zip read them as one stream) -- the text a reader
// sees is the same, and deflate, two thirds of what the C2 job's 20 000 records cost after the kernel, runs in parallel.
void putIbdText(gzFile file, int fd, const std::string& text)
{
  const size_t piece = (size_t)128 << 10;
  const size_t nThreads = std::min<size_t>(outputThreads(), text.size() / piece);
  if (fd < 0 || nThreads < 2) {
    for (size_t off = 0; off < text.size(); off += (size_t)1 << 30) {
      gzwrite(file, text.data() + off, static_cast<unsigned>(std::min<size_t>(text.size() - off, (size_t)1 << 30)));
    }
    return;
  }
  std::vector<size_t> cut(nThreads + 1, text.size());
  cut[0] = 0;
  for (size_t t = 1; t < nThreads; ++t) {
    const size_t at = text.find('\n', text.size() * t / nThreads);
    cut[t] = at == std::string::npos ? text.size() : at + 1;
  }
  std::vector<std::future<std::string>> members;
  for (size_t t = 0; t < nThreads; ++t) {
    members.push_back(std::async(std::launch::async, [&text, &cut, t] {
      return cut[t + 1] > cut[t] ? gzipMember(text.data() + cut[t], cut[t + 1] - cut[t]) : std::string();
    }));
  }
  gzflush(file, Z_FINISH); // the member under way ends here; a later gzwrite starts a new one
  for (auto& m : members) {
    writeAll(fd, m.get());
  }
}
} // namespace

void HMM::closePerPairFiles()
{
  if (mMeanFile) {
    gzclose(mMeanFile);
    mMeanFile = nullptr;
    mMeanFd = -1;
  }
  if (mMapFile) {
    gzclose(mMapFile);
    mMapFile = nullptr;
    mMapFd = -1;
  }
}

void HMM::updateOutputStructures()
{
  // (the reference resizes its per-batch mean / MAP buffers here; the device path has none)
  resetDecoding(); // HMM.cpp:1757
}

void HMM::setStorePerPairPosteriorMean(bool v)
{
  flush(); // (what is queued was decoded under the old setting in the reference)
  mStoreMean = v;
  updateOutputStructures();
}

void HMM::setStorePerPairMap(bool v)
{
  flush(); // (what is queued was decoded under the old setting in the reference)
  mStoreMap = v;
  updateOutputStructures();
}

void HMM::setStorePerPairPosterior(bool v)
{
  flush(); // (what is queued was decoded under the old setting in the reference)
  mStorePosterior = v;
  updateOutputStructures();
}

void HMM::setStoreSumOfPosterior(bool v)
{
  flush(); // (what is queued was decoded under the old setting in the reference)
  mStoreSumOfPosterior = v;
  updateOutputStructures();
}

void HMM::setWritePerPairPosteriorMean(bool v)
{
  flush(); // (what is queued was decoded under the old setting in the reference)
  mWriteMean = v;
  updateOutputStructures();
}

void HMM::setWritePerPairMap(bool v)
{
  flush(); // (what is queued was decoded under the old setting in the reference)
  mWriteMap = v;
  updateOutputStructures();
}

void HMM::resetDecoding()
{
  // HMM.cpp:259-271: the per-pair text outputs of ASMC mode are (re)opened here
  closePerPairFiles();
  mPerPairRows = 0;
  auto openOut = [&](const std::string& suffix, int& fd) {
    const std::string name = mParams.outFileRoot + suffix;
    gzFile f = gzOpenWithFd(name, "w", fd);
    if (!f) {
      throw std::runtime_error("ERROR: could not open " + name);
    }
    return f;
  };
  if (mWriteMean && !mParams.FastSMC) {
    mMeanFile = openOut(".perPairPosteriorMeans.gz", mMeanFd);
  }
  if (mWriteMap && !mParams.FastSMC) {
    mMapFile = openOut(".perPairMAP.gz", mMapFd);
  }
  const size_t n = static_cast<size_t>(mData.sites) * mDq.states;
  mReturn.sumOverPairs.assign(n, 0.f);
  if (mParams.doMajorMinorPosteriorSums) {
    mReturn.sumOverPairs00.assign(n, 0.f);
    mReturn.sumOverPairs01.assign(n, 0.f);
    mReturn.sumOverPairs11.assign(n, 0.f);
  }
}

// The first HIP call of a process initialises the runtime (0.15 s on the GPU box: most of what FastSMC.run() spent
// before its kernel was in flight).  The drivers start it on a helper thread BEFORE they read their input files -- a
// context is created and destroyed, which leaves the runtime up -- and the first real engine() waits for that thread.
// Without a device the attempt fails quietly; the error surfaces where the engine is actually needed.
namespace
{
std::mutex gWarmMutex;
std::shared_future<void> gWarm;
} // namespace

void warmUpDevice(int device)
{
  std::lock_guard<std::mutex> lock(gWarmMutex);
  if (gWarm.valid()) {
    return;
  }
  gWarm = std::async(std::launch::async, [device] {
            fsmc_ctx* c = nullptr;
            if (fsmc_ctx_create(device, nullptr, &c) == FSMC_OK) {
              fsmc_ctx_destroy(c);
            }
          }).share();
}

fsmc_ctx* HMM::engine()
{
  if (!mCtx) {
    {
      std::shared_future<void> warm;
      {
        std::lock_guard<std::mutex> lock(gWarmMutex);
        warm = gWarm;
      }
      if (warm.valid()) {
        warm.wait();
      }
    }
    int rc = fsmc_ctx_create(mParams.gpuDevice, nullptr, &mCtx);
    if (rc != FSMC_OK) {
      mCtx = nullptr;
      throw std::runtime_error(std::string("cannot open the MI355X decode engine: ") + fsmc_last_error(nullptr));
    }
  }
  return mCtx;
}

namespace
{
void putIbdText(gzFile file, int fd, const std::string& text); // (below, beside openIbdFile)
// FSMC_HOST_TIMING: wall-clock marks of a job's phases on stderr (seconds since the first mark of the process)
void hostMark(const char* what)
{
  static const bool on = std::getenv("FSMC_HOST_TIMING") != nullptr;
  if (!on) {
    return;
  }
  using Clock = std::chrono::steady_clock;
  static const Clock::time_point t0 = Clock::now();
  std::fprintf(stderr, "[fsmc host] %8.3f s  %s\n", std::chrono::duration<double>(Clock::now() - t0).count(), what);
}
} // namespace

void HMM::ensureEngine()
{
  hostMark("engine: open");
  engine();
  if (!mModel) {
    fsmc_model_desc d{};
    d.K = mPrep.K;
    d.S = mPrep.S;
    d.pi = mPrep.pi.data();
    d.col_ratios = mPrep.colRatios.data();
    d.exp_times = mPrep.expTimes.data();
    d.n_rows = mPrep.nRows;
    d.D = mPrep.D.data();
    d.B = mPrep.B.data();
    d.U = mPrep.U.data();
    d.RR = mPrep.RR.data();
    d.step_row = mPrep.stepRow.data();
    d.e1 = mPrep.e1.data();
    d.e0m1 = mPrep.e0m1.data();
    d.e2m0 = mPrep.e2m0.data();
    d.state_threshold = mPrep.stateThreshold;
    d.age_threshold = mPrep.ageThreshold;
    d.probability_threshold = mPrep.probabilityThreshold;
    if (mPrep.sequence) {
      d.sequence = 1;
      d.gap_row_f = mPrep.gapRowF.data();
      d.site_row_f = mPrep.siteRowF.data();
      d.gap_row_b = mPrep.gapRowB.data();
      d.site_row_b = mPrep.siteRowB.data();
      d.hom = mPrep.hom.data();
    }
    check(mCtx, fsmc_model_create(mCtx, &d, &mModel), "fsmc_model_create");
    hostMark("engine: model on the device");
  }
  if (!mHapsUploaded) {
    check(mCtx,
          fsmc_haps_upload(mCtx, mData.bits.data(), static_cast<uint32_t>(mData.numHapRows()),
                           static_cast<uint32_t>(mData.sites)),
          "fsmc_haps_upload");
    mHapsUploaded = true;
    hostMark("engine: haplotypes on the device");
  }
}

// ------------------------------------------------------------------ queueing (HMM.cpp:283-636)

PairObservations HMM::makePairObs(int_least8_t iHap, unsigned ind1, int_least8_t jHap, unsigned ind2) const
{
  PairObservations ret;
  ret.iHap = iHap;
  ret.jHap = jHap;
  ret.iInd = ind1;
  ret.jInd = ind2;
  const bool wholeSequence = !(mParams.FastSMC && mParams.hashing);
  if (wholeSequence) { // HMM.cpp:139-142; in hashing mode bits are made per batch window
    const size_t ra = dipToHapId(ind1, static_cast<unsigned long>(iHap));
    const size_t rb = dipToHapId(ind2, static_cast<unsigned long>(jHap));
    const size_t S = static_cast<size_t>(mData.sites);
    ret.obsBits.resize(S);
    ret.homMinorBits.resize(S);
    for (size_t s = 0; s < S; ++s) {
      const bool a = mData.genotype(ra, s), b = mData.genotype(rb, s);
      ret.obsBits[s] = a ^ b;
      ret.homMinorBits[s] = a & b;
    }
  }
  return ret;
}

std::vector<PairObservations> HMM::getBatchBuffer() const
{
  std::vector<PairObservations> out;
  for (size_t i = mBatchBegin; i < mPairs.size(); ++i) {
    const auto [indA, hapA] = hapToDipId(mPairs[i].hap_a);
    const auto [indB, hapB] = hapToDipId(mPairs[i].hap_b);
    out.push_back(makePairObs(static_cast<int_least8_t>(hapA), static_cast<unsigned>(indA),
                              static_cast<int_least8_t>(hapB), static_cast<unsigned>(indB)));
  }
  return out;
}

void HMM::queuePair(unsigned hapRowA, unsigned hapRowB)
{
  if (hapRowA >= mData.numHapRows() || hapRowB >= mData.numHapRows()) {
    throw std::runtime_error("haplotype index out of range");
  }
  mPairs.push_back(fsmc_pair{hapRowA, hapRowB});
  if (mPairs.size() - mBatchBegin == static_cast<size_t>(mBatchSize)) {
    closeBatch(false);
    if (mPairs.size() >= mFlushThreshold) {
      flush();
    }
  }
}

void HMM::closeBatch(bool last)
{
  // addToBatch / runLastBatch (HMM.cpp:555-636): the batch's decode window is the union of its pairs'
  // windows padded by 0.5 cM; in hashing mode every pair is scanned over the un-padded union
  // (HMM.cpp:1199-1204).  The reference pads the last batch to a multiple of its SIMD width with
  // copies of the last pair; lanes are independent, so no padding is needed here.
  const size_t n = mPairs.size() - mBatchBegin;
  if (n == 0) {
    return;
  }
  const size_t slots = last ? n : static_cast<size_t>(mBatchSize);
  const unsigned startBatch = *std::min_element(mFromBatch.begin(), mFromBatch.begin() + slots);
  const unsigned endBatch = *std::max_element(mToBatch.begin(), mToBatch.begin() + slots);
  const unsigned from = getFromPosition(mData.geneticPositions, startBatch);
  const unsigned to = getToPosition(mData.geneticPositions, endBatch);
  unsigned scanFrom = 0, scanTo = static_cast<unsigned>(mData.sites);
  if (mParams.FastSMC && mParams.hashing) {
    scanFrom = startBatch;
    scanTo = endBatch;
  }
  if (scanTo <= scanFrom) {
    // the reference's scan loop would be empty: nothing is ever output for this batch
    mPairs.resize(mBatchBegin);
    return;
  }
  // Lane packing.  A wavefront decodes a group of up to 64 pairs that share a decode window and a scan window, and
  // nothing a pair's output depends on involves the other pairs of its group.  So a batch smaller than a wavefront
  // (the FastSMC default is 32) is appended to the previous group while that has free lanes and the SAME windows:
  // same records, same order (they are ordered by pair), fuller waves.  Outside hashing mode every window is the
  // whole sequence and every group fills up; in hashing mode consecutive batches closed at the same word often share
  // their union window.  Not for the sums over pairs -- there the reference adds batch by batch (HMM.cpp:1054-1073)
  // and a group stays one batch.
  const bool packLanes = !mParams.doPosteriorSums && !mParams.doMajorMinorPosteriorSums;
  size_t off = 0;
  if (packLanes && !mGroups.empty()) {
    fsmc_group& g = mGroups.back();
    if (g.first_pair + g.n_pairs == mBatchBegin && g.n_pairs < 64 && g.from == from && g.to == to &&
        g.scan_from == scanFrom && g.scan_to == scanTo) {
      const size_t take = std::min<size_t>(64 - g.n_pairs, n);
      g.n_pairs += static_cast<uint32_t>(take);
      off = take;
    }
  }
  if (!packLanes) {
    mBatchFirstGroup.push_back(static_cast<uint32_t>(mGroups.size()));
  }
  for (; off < n; off += 64) {
    fsmc_group g{};
    g.first_pair = static_cast<uint32_t>(mBatchBegin + off);
    g.n_pairs = static_cast<uint32_t>(std::min<size_t>(64, n - off));
    g.from = from;
    g.to = to;
    g.scan_from = scanFrom;
    g.scan_to = scanTo;
    mGroups.push_back(g);
  }
  mBatchBegin = mPairs.size();
}

void HMM::flush()
{
  if (mGroups.empty()) {
    return;
  }
  ensureEngine();
  const size_t nPairs = mBatchBegin; // pairs covered by closed batches
  using Clock = std::chrono::steady_clock;
  auto since = [](const Clock::time_point t0) { return std::chrono::duration<double>(Clock::now() - t0).count(); };
  Clock::time_point t0 = Clock::now();
  check(mCtx, fsmc_worklist_upload(mCtx, mPairs.data(), nPairs, mGroups.data(), mGroups.size()),
        "fsmc_worklist_upload");
  mTimeUpload += since(t0);
  hostMark("flush: work list uploaded");

  if (mParams.FastSMC) {
    uint32_t flags = 0;
    if (mParams.doPerPairPosteriorMean) flags |= FSMC_WANT_MEAN;
    if (mParams.doPerPairMAP) flags |= FSMC_WANT_MAP;
    t0 = Clock::now();
    check(mCtx, fsmc_decode_ibd_launch(mCtx, mModel, flags), "fsmc_decode_ibd_launch");
    hostMark("flush: kernel launched");
    std::vector<fsmc_ibd_record> recs(std::max<size_t>(1024, 4 * nPairs));
    size_t n = 0;
    int rc = fsmc_decode_ibd_fetch(mCtx, recs.data(), recs.size(), &n);
    if (rc == FSMC_EOVERFLOW) {
      recs.resize(n);
      rc = fsmc_decode_ibd_fetch(mCtx, recs.data(), recs.size(), &n);
    }
    check(mCtx, rc, "fsmc_decode_ibd_fetch");
    mTimeDecode += since(t0);
    hostMark("flush: records fetched");
    t0 = Clock::now();
    if (mIbdFile && !mParams.BIN_OUT && n >= 512) {
      // a flush's text in one piece (formatted by several threads), then what writeIbd does beside the text
      std::vector<uint32_t> pairOf(n);
      for (size_t i = 0; i < n; ++i) {
        pairOf[i] = recs[i].pair;
      }
      putIbdText(mIbdFile, mIbdFd, formatIbdRecords(mPairs.data(), recs.data(), n, pairOf.data()));
      mSegmentsDetected += n;
      if (mKeepRecords) {
        for (size_t i = 0; i < n; ++i) {
          mKeptOrdinals.push_back(mPairsFlushed + recs[i].pair);
          mKeptRecords.push_back(recs[i]);
          mKeptPairs.push_back(mPairs[recs[i].pair]);
        }
      }
    } else {
      for (size_t i = 0; i < n; ++i) {
        if (mKeepRecords) {
          mKeptOrdinals.push_back(mPairsFlushed + recs[i].pair);
        }
        writeIbd(mPairs[recs[i].pair], recs[i]);
      }
    }
    mTimeWrite += since(t0);
  }
  if (mParams.doPosteriorSums || mParams.doMajorMinorPosteriorSums) {
    const bool mm = mParams.doMajorMinorPosteriorSums;
    // the reference sums a WHOLE batch over its pairs and then adds it (HMM.cpp:1054-1073): the groups of a batch of more
    // than 64 pairs share one running sum on the device
    mBatchFirstGroup.push_back(static_cast<uint32_t>(mGroups.size()));
    check(mCtx,
          fsmc_decode_sums_batches(mCtx, mModel, mBatchFirstGroup.data(), mBatchFirstGroup.size() - 1,
                                   mParams.doPosteriorSums ? mReturn.sumOverPairs.data() : nullptr,
                                   mm ? mReturn.sumOverPairs00.data() : nullptr,
                                   mm ? mReturn.sumOverPairs01.data() : nullptr,
                                   mm ? mReturn.sumOverPairs11.data() : nullptr),
          "fsmc_decode_sums_batches");
  }
  const bool writeFiles = mMeanFile || mMapFile;
  const bool storeAny = mStoreMean || mStoreMap || mStorePosterior || mStoreSumOfPosterior;
  if (!mParams.FastSMC && (storeAny || writeFiles)) {
    // writePerPairOutput (HMM.cpp:1360-1458)
    const size_t S = static_cast<size_t>(mData.sites);
    const size_t K = mDq.states;
    auto& R = mPairsReturn;
    const size_t base = R.numWritten;
    if (storeAny && base + nPairs > static_cast<size_t>(R.numPairs)) {
      throw std::runtime_error("more pairs decoded than the return structure was initialised for");
    }
    const bool wantMean = mStoreMean || mStorePosterior || mStoreSumOfPosterior || mMeanFile;
    std::vector<float> mean(wantMean ? nPairs * S : 0);
    std::vector<int32_t> map(mStoreMap || mStoreMean || mMapFile ? nPairs * S : 0);
    check(mCtx,
          fsmc_decode_per_pair(mCtx, mModel, mExpectedCoalTimes.data(), mean.empty() ? nullptr : mean.data(),
                               map.empty() ? nullptr : map.data()),
          "fsmc_decode_per_pair");
    if (mStorePosterior || mStoreSumOfPosterior) {
      // full posteriors (times expected coalescence time, HMM.cpp:1382-1388) come from the posterior dump
      std::vector<float> dump(static_cast<size_t>(64) * K * S * mGroups.size());
      check(mCtx, fsmc_decode_posteriors(mCtx, mModel, dump.data(), dump.size()), "fsmc_decode_posteriors");
      for (size_t g = 0; g < mGroups.size(); ++g) {
        const float* gp = dump.data() + g * 64 * K * S;
        for (uint32_t v = 0; v < mGroups[g].n_pairs; ++v) {
          const size_t pairIdx = mGroups[g].first_pair + v;
          for (size_t pos = 0; pos < S; ++pos) {
            for (size_t k = 0; k < K; ++k) {
              const float postValue = gp[(pos * K + k) * 64 + v] * mExpectedCoalTimes[k];
              if (mStorePosterior) {
                R.perPairPosteriors[base + pairIdx][k * S + pos] = postValue;
              }
              if (mStoreSumOfPosterior) {
                R.sumOfPosteriors[k * S + pos] += postValue;
              }
            }
          }
        }
      }
    }
    if (writeFiles) {
      // HMM.cpp:1412-1420: `fout << matrix.topRows(actualBatchSize).format(m_eigenOutputFormat)` once per BATCH, with
      // IOFormat(FullPrecision, DontAlignCols, " ", "\n") (HMM.hpp:154): coefficients separated by a blank, rows by a
      // newline -- a separator BETWEEN rows: nothing follows a batch's last row, so the next batch's first row
      // continues that line.  Reproduced as it is (the files are the reference's, quirk included): a newline goes in
      // front of every row that is not the first of its batch.  FullPrecision for float is the stream at
      // NumTraits<float>::digits10() = 6 significant digits (Eigen 3.4, the version the reference's unpinned vcpkg
      // dependency resolves to; general notation = "%.6g"); the MAP matrix is integer.  Eigen master / 5.x print
      // max_digits10 = 9 there: FSMC_EIGEN_FULL_PRECISION_DIGITS=9 in the environment writes the file such a build of
      // the reference writes.
      const auto B = static_cast<uint64_t>(mBatchSize);
      // the rows of pairs [lo, hi) as text (the newline rule above counts rows over the whole file)
      auto rowsText = [&](size_t lo, size_t hi, auto&& cell) {
        std::string text;
        text.reserve((hi - lo) * S * 8);
        char buf[48];
        for (size_t i = lo; i < hi; ++i) {
          if ((mPerPairRows + i) % B != 0) {
            text.push_back('\n');
          }
          for (size_t pos = 0; pos < S; ++pos) {
            if (pos) {
              text.push_back(' ');
            }
            text.append(buf, static_cast<size_t>(cell(buf, sizeof(buf), i * S + pos)));
          }
        }
        return text;
      };
      // A few rows: one thread, gzwrite.  Many (a number per pair and site: 300 million for the reference's example
      // cohort): several threads format AND compress blocks of rows into gzip members of their own, written in order
      // behind whatever gzwrite had under way (putIbdText, above: one stream to every reader of gzip files).
      auto writeRows = [&](gzFile f, int fd, auto&& cell) {
        const size_t rowsPerThread = std::max<size_t>(1, ((size_t)1 << 20) / (S * 8 + 1)); // (about a MiB of text a piece)
        const size_t nThreads = std::min<size_t>(outputThreads(), (nPairs + rowsPerThread - 1) / rowsPerThread);
        if (fd < 0 || nThreads < 2) {
          for (size_t lo = 0; lo < nPairs; lo += rowsPerThread) {
            const std::string text = rowsText(lo, std::min(nPairs, lo + rowsPerThread), cell);
            gzwrite(f, text.data(), static_cast<unsigned>(text.size()));
          }
          return;
        }
        gzflush(f, Z_FINISH);
        for (size_t blockLo = 0; blockLo < nPairs; blockLo += nThreads * rowsPerThread) {
          std::vector<std::future<std::string>> members;
          for (size_t t = 0; t < nThreads; ++t) {
            const size_t lo = std::min(nPairs, blockLo + t * rowsPerThread), hi = std::min(nPairs, lo + rowsPerThread);
            if (lo < hi) {
              members.push_back(std::async(std::launch::async, [&, lo, hi] {
                const std::string text = rowsText(lo, hi, cell);
                return gzipMember(text.data(), text.size());
              }));
            }
          }
          for (auto& m : members) {
            writeAll(fd, m.get());
          }
        }
      };
      if (mMeanFile) {
        int digits = 6;
        if (const char* v = std::getenv("FSMC_EIGEN_FULL_PRECISION_DIGITS")) {
          const int d = std::atoi(v);
          if (d != 6 && d != 9) {
            throw std::runtime_error("FSMC_EIGEN_FULL_PRECISION_DIGITS is 6 (Eigen 3.4) or 9 (Eigen 5)");
          }
          digits = d;
        }
        writeRows(mMeanFile, mMeanFd, [&](char* b, size_t n, size_t idx) {
          return std::snprintf(b, n, "%.*g", digits, static_cast<double>(mean[idx]));
        });
      }
      if (mMapFile) {
        writeRows(mMapFile, mMapFd, [&](char* b, size_t n, size_t idx) { return std::snprintf(b, n, "%d", map[idx]); });
      }
      mPerPairRows += nPairs;
    }
    for (size_t i = 0; storeAny && i < nPairs; ++i) {
      const auto [indA, hapA] = hapToDipId(mPairs[i].hap_a);
      const auto [indB, hapB] = hapToDipId(mPairs[i].hap_b);
      R.perPairIndices.at(base + i) =
          std::make_tuple(static_cast<unsigned long>(mPairs[i].hap_a), indPlusHapToCombinedId(mData.IIDList.at(indA), hapA),
                          static_cast<unsigned long>(mPairs[i].hap_b), indPlusHapToCombinedId(mData.IIDList.at(indB), hapB));
      if (mStoreMean) {
        std::copy(mean.begin() + i * S, mean.begin() + (i + 1) * S, R.perPairPosteriorMeans.begin() + (base + i) * S);
        // the reference stores the MAP rows under the posterior-mean flag (HMM.cpp:1447-1449)
        if (!R.perPairMAPs.empty() && !map.empty()) {
          std::copy(map.begin() + i * S, map.begin() + (i + 1) * S, R.perPairMAPs.begin() + (base + i) * S);
        }
      }
    }
    if (storeAny) {
      R.numWritten += nPairs;
    }
  }

  // keep any pairs of a still-open batch
  std::vector<fsmc_pair> rest(mPairs.begin() + static_cast<long>(mBatchBegin), mPairs.end());
  mPairs.swap(rest);
  mPairsFlushed += nPairs;
  mGroups.clear();
  mBatchFirstGroup.clear();
  mBatchBegin = 0;
}

void HMM::setShard(int rank, int world)
{
  if (world < 1 || rank < 0 || rank >= world) {
    throw std::runtime_error("setShard: need 0 <= rank < world");
  }
  mShardRank = rank;
  mShardWorld = world;
}

std::pair<unsigned long long, unsigned long long> HMM::shardBatchRange(unsigned long long nBatches) const
{
  const auto r = static_cast<unsigned long long>(mShardRank), w = static_cast<unsigned long long>(mShardWorld);
  return {nBatches * r / w, nBatches * (r + 1) / w};
}

std::string HMM::ibdFileName(int jobs, int jobInd) const
{
  std::string name = mParams.outFileRoot + "." + std::to_string(jobInd) + "." + std::to_string(jobs) +
                     (mParams.BIN_OUT ? ".FastSMC.bibd.gz" : ".FastSMC.ibd.gz");
  if (mShardWorld > 1) {
    name += ".part" + std::to_string(mShardRank) + "of" + std::to_string(mShardWorld);
  }
  return name;
}

void HMM::openIbdFile(int jobs, int jobInd)
{
  if (mIbdFile) {
    gzclose(mIbdFile);
    mIbdFile = nullptr;
    mIbdFd = -1;
  }
  if (!mWriteIbdFile) {
    return; // (records are kept and gathered: setWriteIbdFile)
  }
  const std::string name = ibdFileName(jobs, jobInd);
  mIbdFile = gzOpenWithFd(name, mParams.BIN_OUT ? "wb" : "w", mIbdFd);
  if (!mIbdFile) {
    throw std::runtime_error("cannot open IBD output file " + name);
  }
  if (mParams.BIN_OUT && mShardRank == 0) {
    writeBinaryHeader(mIbdFile);
  }
}

void HMM::writeIbdRecordsTo(const std::string& fileName, const std::vector<fsmc_pair>& pairs,
                            const std::vector<fsmc_ibd_record>& records) const
{
  if (pairs.size() != records.size()) {
    throw std::runtime_error("writeIbdRecordsTo: one pair per record");
  }
  const int S = static_cast<int>(mData.sites);
  for (size_t i = 0; i < records.size(); ++i) {
    const fsmc_ibd_record& r = records[i];
    if (r.start < 0 || r.end < r.start || r.end >= S || pairs[i].hap_a >= 2 * mData.numIndividuals() ||
        pairs[i].hap_b >= 2 * mData.numIndividuals()) {
      throw std::runtime_error("writeIbdRecordsTo: record " + std::to_string(i) + " lies outside this data set");
    }
  }
  int fd = -1;
  gzFile file = gzOpenWithFd(fileName, mParams.BIN_OUT ? "wb" : "w", fd);
  if (!file) {
    throw std::runtime_error("cannot open IBD output file " + fileName);
  }
  if (mParams.BIN_OUT) {
    writeBinaryHeader(file);
  }
  if (!mParams.BIN_OUT) {
    putIbdText(file, fd, formatIbdRecords(pairs.data(), records.data(), records.size()));
  } else {
    for (size_t i = 0; i < records.size(); ++i) {
      emitIbd(file, pairs[i], records[i]);
    }
  }
  gzclose(file);
}

std::pair<unsigned long long, unsigned long long> HMM::pairRangeOfJob(int jobs, int jobInd, bool shardOnly) const
{
  const unsigned long long N = mData.numIndividuals();
  // pair range of this job (HMM.cpp:310-321)
  const unsigned long long totPairs = mParams.withinOnly ? N : 2 * N * N - N;
  const unsigned long long pairsStart = totPairs * static_cast<unsigned long long>(jobInd - 1) / jobs;
  const unsigned long long pairsEnd = totPairs * static_cast<unsigned long long>(jobInd) / jobs;
  unsigned long long lo = pairsStart, hi = pairsEnd;
  if (shardOnly) {
    // this device's share of the job: whole batches, contiguous (setShard)
    const auto B = static_cast<unsigned long long>(mBatchSize);
    const auto [batchLo, batchHi] = shardBatchRange((pairsEnd - pairsStart + B - 1) / B);
    lo = std::min(pairsEnd, pairsStart + batchLo * B);
    hi = std::min(pairsEnd, pairsStart + batchHi * B);
  }
  return {lo, hi};
}

template <typename Fn> void HMM::forEachPairOfJob(int jobs, int jobInd, bool shardOnly, Fn&& fn) const
{
  const unsigned long long N = mData.numIndividuals();
  const auto [lo, hi] = pairRangeOfJob(jobs, jobInd, shardOnly);
  // individual i contributes the 4 * i cross pairs with every j < i and then its own two haplotypes: its block starts
  // at ordinal 2 * i * i - i (HMM.cpp:325-357); blocks outside [lo, hi) are skipped whole
  unsigned long long pairs = 0;
  for (unsigned long long i = 0; i < N && pairs < hi; i++) {
    const unsigned long long block = mParams.withinOnly ? 1ull : 4ull * i + 1ull;
    if (pairs + block <= lo) {
      pairs += block;
      continue;
    }
    if (!mParams.withinOnly) {
      for (unsigned long long j = 0; j < i; j++) {
        if (pairs + 4 <= lo || pairs >= hi) {
          pairs += 4;
          continue;
        }
        for (int iHap = 1; iHap <= 2; iHap++) {
          for (int jHap = 1; jHap <= 2; jHap++) {
            if (lo <= pairs && pairs < hi) {
              // makePairObs(jHap, j, iHap, i): the lower-numbered individual is the record's first id
              fn(static_cast<unsigned>(2 * j + jHap - 1), static_cast<unsigned>(2 * i + iHap - 1));
            }
            pairs++;
          }
        }
      }
    }
    if (lo <= pairs && pairs < hi) {
      fn(static_cast<unsigned>(2 * i), static_cast<unsigned>(2 * i + 1));
    }
    pairs++;
  }
}

std::vector<std::pair<unsigned, unsigned>> HMM::pairsOfJob(int jobs, int jobInd) const
{
  std::vector<std::pair<unsigned, unsigned>> out;
  forEachPairOfJob(jobs, jobInd, false, [&](unsigned a, unsigned b) { out.emplace_back(a, b); });
  return out;
}

void HMM::decodeAll(int jobs, int jobInd)
{
  hostMark("decodeAll: start");
  resetDecoding();
  if (mParams.FastSMC) {
    openIbdFile(jobs, jobInd);
    if (mParams.hashing) {
      return; // pairs arrive through decodeFromHashing
    }
  }
  // the job's size is known here (its pair range, HMM.cpp:310-321): announce it, so that a long job under the
  // library's own workspace policy allocates once at its start instead of growing into the card (fsmc_ctx_expect_work)
  const auto [pairLo, pairHi] = pairRangeOfJob(jobs, jobInd, true);
  const unsigned long long nPairsOfJob = pairHi - pairLo;
  if (nPairsOfJob) {
    announceWork(static_cast<double>(nPairsOfJob) * static_cast<double>(mData.sites));
    // (the work list of a flush: at most the flush threshold plus a batch)
    const size_t expect = static_cast<size_t>(std::min<unsigned long long>(nPairsOfJob, mFlushThreshold + 64));
    mPairs.reserve(mPairs.size() + expect);
    mGroups.reserve(mGroups.size() + expect / 32 + 2);
  }
  forEachPairOfJob(jobs, jobInd, true, [&](unsigned a, unsigned b) { queuePair(a, b); });
  hostMark("decodeAll: pairs queued");
  finishDecoding();
  hostMark("decodeAll: done");
}

void HMM::announceWork(double pairSites)
{
  // every consumer of a flush is a launch over the same pairs (IBD decode; sums over pairs)
  int launches = 0;
  if (mParams.FastSMC) launches++;
  if (mParams.doPosteriorSums || mParams.doMajorMinorPosteriorSums) launches++;
  if (launches == 0) launches = 1;
  check(engine(), fsmc_ctx_expect_work(engine(), pairSites * launches, static_cast<int32_t>(mDq.states)),
        "fsmc_ctx_expect_work");
}

void HMM::decodePairs(const std::vector<unsigned>& A, const std::vector<unsigned>& B)
{
  if (A.size() != B.size()) {
    throw std::runtime_error("vector of A indicies must be the same size as vector of B indicies");
  }
  for (size_t i = 0; i < A.size(); ++i) {
    decodePair(A[i], B[i]);
  }
}

void HMM::decodePair(const unsigned i, const unsigned j)
{
  if (i >= mData.numIndividuals() || j >= mData.numIndividuals()) {
    throw std::runtime_error("individual index out of range");
  }
  if (i != j) {
    for (int iHap = 1; iHap <= 2; iHap++) {
      for (int jHap = 1; jHap <= 2; jHap++) {
        queuePair(static_cast<unsigned>(dipToHapId(i, iHap)), static_cast<unsigned>(dipToHapId(j, jHap)));
      }
    }
  } else {
    queuePair(static_cast<unsigned>(dipToHapId(i, 1)), static_cast<unsigned>(dipToHapId(i, 2)));
  }
}

void HMM::decodeHapPair(const unsigned long i, const unsigned long j)
{
  queuePair(static_cast<unsigned>(i), static_cast<unsigned>(j));
}

void HMM::decodeHapPairs(const std::vector<unsigned long>& A, const std::vector<unsigned long>& B)
{
  if (A.size() != B.size()) {
    throw std::runtime_error("vector of A indices must be the same size as vector of B indices");
  }
  for (size_t i = 0; i < A.size(); ++i) {
    decodeHapPair(A[i], B[i]);
  }
}

void HMM::decodeFromHashing(const unsigned hapA, const unsigned hapB, const unsigned fromPos, const unsigned toPos)
{
  if (hapA / 2 >= mData.numIndividuals() || hapB / 2 >= mData.numIndividuals() ||
      fromPos >= static_cast<unsigned>(mData.sites) || toPos >= static_cast<unsigned>(mData.sites)) {
    throw std::runtime_error("decodeFromHashing: index out of range");
  }
  const size_t slot = mHashingCount % static_cast<unsigned long>(mBatchSize);
  mFromBatch[slot] = fromPos;
  mToBatch[slot] = toPos;
  mHashingCount++;
  queuePair(hapA, hapB); // hap = id % 2 == 0 ? 1 : 2, ind = id / 2 (HMM.cpp:483-486): the row itself
}

void HMM::finishDecoding()
{
  closeBatch(true);
  flush();
  if (mCtx) { // the announced job (decodeAll, the hashing driver) is over: no phantom work stays behind
    fsmc_ctx_expect_work(mCtx, 0.0, static_cast<int32_t>(mDq.states));
  }
  closePerPairFiles(); // HMM.cpp:518-523
  if (!(mParams.FastSMC && mParams.hashing)) {
    std::fill(mFromBatch.begin(), mFromBatch.end(), 0u);
    std::fill(mToBatch.begin(), mToBatch.end(), static_cast<unsigned>(mData.sites));
  }
}

void HMM::finishFromHashing()
{
  closeBatch(true);
  flush();
  if (mCtx) {
    fsmc_ctx_expect_work(mCtx, 0.0, static_cast<int32_t>(mDq.states));
  }
  closeIBDFile();
  if (std::getenv("FSMC_HOST_TIMING")) {
    std::fprintf(stderr, "[fsmc host] work-list upload %.3f s, decode (launch + fetch) %.3f s, records out %.3f s\n",
                 mTimeUpload, mTimeDecode, mTimeWrite);
  }
}

void HMM::closeIBDFile()
{
  if (mIbdFile) {
    gzclose(mIbdFile);
    mIbdFile = nullptr;
    mIbdFd = -1;
  }
}

// ------------------------------------------------------------------ output (HMM.cpp:1110-1177, 383-401)

std::string HMM::formatIbdRecord(const fsmc_pair& pr, const fsmc_ibd_record& r) const
{
  std::stringstream record;
  record << std::setprecision(std::numeric_limits<float>::digits10 + 1);
  putIbdRecord(record, pr, r);
  return record.str();
}

std::string HMM::formatIbdRecords(const fsmc_pair* pairs, const fsmc_ibd_record* records, size_t n,
                                  const uint32_t* pairOf) const
{
  auto part = [&](size_t lo, size_t hi) {
    std::ostringstream text;
    text << std::setprecision(std::numeric_limits<float>::digits10 + 1);
    for (size_t i = lo; i < hi; ++i) {
      putIbdRecord(text, pairs[pairOf ? pairOf[i] : i], records[i]);
    }
    return text.str();
  };
  const size_t perThread = 256;
  const size_t nThreads = std::min<size_t>(outputThreads(), n / perThread);
  if (nThreads < 2) {
    return part(0, n);
  }
  std::vector<std::future<std::string>> parts;
  for (size_t t = 0; t < nThreads; ++t) {
    parts.push_back(std::async(std::launch::async, part, n * t / nThreads, n * (t + 1) / nThreads));
  }
  std::string out;
  for (auto& f : parts) {
    out += f.get();
  }
  return out;
}

// (the stream carries the precision: std::numeric_limits<float>::digits10 + 1)
void HMM::putIbdRecord(std::ostream& record, const fsmc_pair& pr, const fsmc_ibd_record& r) const
{
  const auto [iInd, iHap] = hapToDipId(pr.hap_a);
  const auto [jInd, jHap] = hapToDipId(pr.hap_b);
  record << mData.FamIDList[iInd] << '\t' << mData.IIDList[iInd] << '\t' << static_cast<int>(iHap) << '\t'
         << mData.FamIDList[jInd] << '\t' << mData.IIDList[jInd] << '\t' << static_cast<int>(jHap) << '\t'
         << mData.chrNumber;
  record << '\t' << mData.physicalPositions[r.start] << '\t' << mData.physicalPositions[r.end];
  if (mParams.outputIbdSegmentLength) {
    const float length_cM = 100.f * (mData.geneticPositions[r.end] - mData.geneticPositions[r.start]);
    record << '\t' << length_cM;
  }
  const double ibd_score = r.prob / static_cast<double>(static_cast<unsigned>(r.end - r.start) + 1u);
  record << '\t' << ibd_score;
  if (mParams.doPerPairPosteriorMean) {
    record << '\t' << r.post_mean;
  }
  if (mParams.doPerPairMAP) {
    record << '\t' << r.map;
  }
  record << '\n';
}

void HMM::writeIbd(const fsmc_pair& pr, const fsmc_ibd_record& r)
{
  mSegmentsDetected++;
  if (mKeepRecords) {
    mKeptRecords.push_back(r);
    mKeptPairs.push_back(pr);
  }
  if (!mIbdFile) {
    return;
  }
  emitIbd(mIbdFile, pr, r);
}

void HMM::emitIbd(gzFile file, const fsmc_pair& pr, const fsmc_ibd_record& r) const
{
  if (!mParams.BIN_OUT) {
    const std::string s = formatIbdRecord(pr, r);
    gzwrite(file, s.c_str(), static_cast<unsigned>(s.size()));
    return;
  }
  const auto [iInd, iHap] = hapToDipId(pr.hap_a);
  const auto [jInd, jHap] = hapToDipId(pr.hap_b);
  const unsigned int ind[2] = {static_cast<unsigned>(iInd), static_cast<unsigned>(jInd)};
  const std::uint_least8_t hap[2] = {static_cast<std::uint_least8_t>(iHap), static_cast<std::uint_least8_t>(jHap)};
  const int pos[2] = {mData.physicalPositions[r.start], mData.physicalPositions[r.end]};
  const float ibd_score =
      static_cast<float>(r.prob / static_cast<double>(static_cast<unsigned>(r.end - r.start) + 1u));
  gzwrite(file, &ind[0], sizeof(unsigned int));
  gzwrite(file, &hap[0], sizeof(std::uint_least8_t));
  gzwrite(file, &ind[1], sizeof(unsigned int));
  gzwrite(file, &hap[1], sizeof(std::uint_least8_t));
  gzwrite(file, &pos[0], sizeof(int));
  gzwrite(file, &pos[1], sizeof(int));
  if (mParams.outputIbdSegmentLength) {
    const float length_cM = 100.f * (mData.geneticPositions[r.end] - mData.geneticPositions[r.start]);
    gzwrite(file, &length_cM, sizeof(float));
  }
  gzwrite(file, &ibd_score, sizeof(float));
  if (mParams.doPerPairPosteriorMean) {
    gzwrite(file, &r.post_mean, sizeof(float));
  }
  if (mParams.doPerPairMAP) {
    gzwrite(file, &r.map, sizeof(float));
  }
}

void HMM::writeBinaryHeader(gzFile file) const
{
  gzwrite(file, &mParams.outputIbdSegmentLength, sizeof(bool));
  gzwrite(file, &mParams.doPerPairPosteriorMean, sizeof(bool));
  gzwrite(file, &mParams.doPerPairMAP, sizeof(bool));
  gzwrite(file, &mData.chrNumber, sizeof(int));
  const unsigned int nbInd = static_cast<unsigned>(mData.numIndividuals());
  gzwrite(file, &nbInd, sizeof(unsigned int));
  for (unsigned i = 0; i < nbInd; i++) {
    const unsigned lengthFamid = static_cast<unsigned>(mData.FamIDList[i].size());
    gzwrite(file, &lengthFamid, sizeof(unsigned int));
    gzwrite(file, mData.FamIDList[i].c_str(), lengthFamid);
    const unsigned lengthIid = static_cast<unsigned>(mData.IIDList[i].size());
    gzwrite(file, &lengthIid, sizeof(unsigned int));
    gzwrite(file, mData.IIDList[i].c_str(), lengthIid);
  }
}

// ------------------------------------------------------------------ single-pair decode

std::vector<std::vector<float>> HMM::decode(const PairObservations& obs)
{
  return decode(obs, 0, static_cast<unsigned>(mData.sites));
}

std::vector<std::vector<float>> HMM::decode(const PairObservations& obs, unsigned from, unsigned to)
{
  if (!(from < to) || to > static_cast<unsigned>(mData.sites)) {
    throw std::runtime_error("decode: need from < to <= sites");
  }
  if (!mGroups.empty() || !mPairs.empty()) {
    throw std::runtime_error("decode: pairs are queued; call finishDecoding() first");
  }
  ensureEngine();
  const fsmc_pair pr{static_cast<uint32_t>(dipToHapId(obs.iInd, static_cast<unsigned long>(obs.iHap))),
                     static_cast<uint32_t>(dipToHapId(obs.jInd, static_cast<unsigned long>(obs.jHap)))};
  const fsmc_group g{0, 1, from, to, from, to};
  check(mCtx, fsmc_worklist_upload(mCtx, &pr, 1, &g, 1), "fsmc_worklist_upload");
  const size_t K = mDq.states;
  std::vector<float> dump(static_cast<size_t>(64) * K * (to - from));
  check(mCtx, fsmc_decode_posteriors(mCtx, mModel, dump.data(), dump.size()), "fsmc_decode_posteriors");
  std::vector<std::vector<float>> posterior(K, std::vector<float>(static_cast<size_t>(mData.sites), 0.f));
  for (unsigned pos = from; pos < to; ++pos) {
    for (size_t k = 0; k < K; ++k) {
      posterior[k][pos] = dump[(static_cast<size_t>(pos - from) * K + k) * 64];
    }
  }
  if (mParams.doPosteriorSums) {
    for (size_t k = 0; k < K; k++) {
      for (size_t pos = 0; pos < static_cast<size_t>(mData.sites); pos++) {
        mReturn.sumOverPairs[pos * K + k] += posterior[k][pos];
      }
    }
  }
  return posterior;
}

std::pair<std::vector<float>, std::vector<float>> HMM::decodeSummarize(const PairObservations& obs)
{
  // HMM.cpp:1498-1517: mean[j] = sum_i posterior[i][j] * expectedTimes[i] (i ascending from 0.f), MAP[j] =
  // expectedTimes of the first state with a strictly larger posterior -- the per-pair consumer of the decode
  // kernel evaluates exactly these two in this order (fsmc_decode_per_pair with expectedTimes as the weights).
  if (mParams.doPosteriorSums) {
    (void)decode(obs); // the reference's decodeSummarize goes through decode(), which adds to sumOverPairs
  }
  if (!mGroups.empty() || !mPairs.empty()) {
    throw std::runtime_error("decodeSummarize: pairs are queued; call finishDecoding() first");
  }
  ensureEngine();
  const size_t S = static_cast<size_t>(mData.sites);
  const fsmc_pair pr{static_cast<uint32_t>(dipToHapId(obs.iInd, static_cast<unsigned long>(obs.iHap))),
                     static_cast<uint32_t>(dipToHapId(obs.jInd, static_cast<unsigned long>(obs.jHap)))};
  const fsmc_group g{0, 1, 0, static_cast<uint32_t>(S), 0, static_cast<uint32_t>(S)};
  check(mCtx, fsmc_worklist_upload(mCtx, &pr, 1, &g, 1), "fsmc_worklist_upload");
  std::vector<float> mean(S);
  std::vector<int32_t> arg(S);
  check(mCtx, fsmc_decode_per_pair(mCtx, mModel, mDq.expectedTimes.data(), mean.data(), arg.data()),
        "fsmc_decode_per_pair");
  std::vector<float> map(S);
  for (size_t s = 0; s < S; ++s) {
    map[s] = mDq.expectedTimes[static_cast<size_t>(arg[s])];
  }
  return {map, mean};
}

} // namespace fsmc_host
