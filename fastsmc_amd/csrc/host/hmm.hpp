// hmm.hpp -- host orchestration of the pairwise decode: the counterpart of the reference's class HMM
// (HMM.hpp / HMM.cpp).  Same public surface for the decode path (decodeAll, decodePair(s),
// decodeHapPair(s), decodeFromHashing, finishDecoding, finishFromHashing, makePairObs, return structs);
// different engine: pairs are queued into a work list of <= 64-pair groups and decoded on the GPU
// through the C ABI of include/fastsmc_hip.h -- this file contains no arithmetic of the hot path.
#pragma once

#include <zlib.h>

#include <cstdint>
#include <iosfwd>
#include <memory>
#include <string>
#include <tuple>
#include <vector>

#include "../../../include/fastsmc_hip.h"
#include "data.hpp"
#include "decoding_params.hpp"
#include "decoding_quantities.hpp"

namespace fsmc_host
{

// HMM.hpp:37-46
struct PairObservations {
  int_least8_t iHap = 0, jHap = 0;
  unsigned int iInd = 0, jInd = 0;
  std::vector<bool> obsBits;
  std::vector<bool> homMinorBits;
};

// HMM.hpp:49-66; arrays are [sites][states] row-major (numpy (S, K))
struct DecodingReturnValues {
  std::vector<float> sumOverPairs, sumOverPairs00, sumOverPairs01, sumOverPairs11;
  int sites = 0;
  unsigned int states = 0;
  std::vector<bool> siteWasFlippedDuringFolding;
};

// DecodePairsReturnStruct.hpp:29-124; matrices row-major
struct DecodePairsReturnStruct {
  std::vector<std::tuple<unsigned long, std::string, unsigned long, std::string>> perPairIndices;
  std::vector<std::vector<float>> perPairPosteriors; // per pair [states][sites]
  std::vector<float> sumOfPosteriors;                // [states][sites]
  std::vector<float> perPairPosteriorMeans;          // [pairs][sites]
  std::vector<float> minPosteriorMeans;              // [sites]
  std::vector<int> argminPosteriorMeans;             // [sites]
  std::vector<int> perPairMAPs;                      // [pairs][sites]
  std::vector<int> minMAPs, argminMAPs;              // [sites]
  long numPairs = 0, numSites = 0, numStates = 0;
  bool storeFullPosteriors = false, storeSumOfPosteriors = false, storePerPairPosteriorMeans = false,
       storePerPairMAPs = false;
  size_t numWritten = 0;

  void initialise(const std::vector<unsigned long>& hapsA, const std::vector<unsigned long>& hapsB, long sites,
                  long states, bool fullPosteriors, bool sumOfPost, bool perPairMeans, bool perPairMaps);
  void finaliseCalculations();
};

// The constant inputs of the path as the constructor leaves them (what fsmc_model_create receives).
struct PreparedModel {
  int K = 0, S = 0;
  std::vector<float> pi, colRatios, expTimes;
  std::vector<float> D, B, U, RR; // [nRows][K], only the rows this data set uses
  int nRows = 0;
  std::vector<int32_t> stepRow;        // [S]
  std::vector<float> e1, e0m1, e2m0;   // [S][K]
  unsigned stateThreshold = 0, ageThreshold = 0;
  float probabilityThreshold = 0.f;
  // sequence mode (HMM.cpp:760-770, 915-925): see fsmc_model_desc
  bool sequence = false;
  std::vector<int32_t> gapRowF, siteRowF, gapRowB, siteRowB; // [S]
  std::vector<float> hom;                                    // [S][K]
};

class HMM
{
public:
  HMM(Data data, const DecodingParams& params, int scalingSkip = 1);
  // same, with decoding quantities already in memory (synthetic models) instead of params.decodingQuantFile
  HMM(Data data, DecodingQuantities dq, const DecodingParams& params, int scalingSkip = 1);
  ~HMM();
  HMM(const HMM&) = delete;
  HMM& operator=(const HMM&) = delete;

  void decodeAll(int jobs, int jobInd);                                          // HMM.cpp:283-381
  // the haplotype-row pairs decodeAll(jobs, jobInd) decodes, in its order (tests; no decoding)
  std::vector<std::pair<unsigned, unsigned>> pairsOfJob(int jobs, int jobInd) const;
  void decodePair(unsigned i, unsigned j);                                       // HMM.cpp:413-440
  void decodePairs(const std::vector<unsigned>& A, const std::vector<unsigned>& B); // HMM.cpp:403-411
  void decodeHapPair(unsigned long i, unsigned long j);                          // HMM.cpp:442-458
  void decodeHapPairs(const std::vector<unsigned long>& A, const std::vector<unsigned long>& B);
  void decodeFromHashing(unsigned hapA, unsigned hapB, unsigned fromPos, unsigned toPos); // HMM.cpp:470-502
  void finishDecoding();    // HMM.cpp:515-524
  void finishFromHashing(); // HMM.cpp:531-552
  void closeIBDFile();

  PairObservations makePairObs(int_least8_t iHap, unsigned ind1, int_least8_t jHap, unsigned ind2) const;
  // posterior of one pair, [states][to-from] (HMM::decode, HMM.cpp:1464-1508 -- here via the batched GPU path)
  std::vector<std::vector<float>> decode(const PairObservations& obs);
  std::vector<std::vector<float>> decode(const PairObservations& obs, unsigned from, unsigned to);
  // (MAP, posterior mean) of one pair, one value per site, both in units of expectedTimes (HMM.cpp:1498-1517)
  std::pair<std::vector<float>, std::vector<float>> decodeSummarize(const PairObservations& obs);

  // The reference decodes a batch the moment it is full (addToBatch, HMM.cpp:555-590); here full batches wait in the
  // work list until a flush (many batches per launch), so the getters of results decode what is waiting first.
  DecodingReturnValues& getDecodingReturnValues()
  {
    flush();
    return mReturn;
  }
  DecodePairsReturnStruct& getDecodePairsReturnStruct()
  {
    flush();
    return mPairsReturn;
  }
  const DecodingQuantities& getDecodingQuantities() const { return mDq; }
  const Data& getData() const { return mData; }
  const PreparedModel& getPreparedModel() const { return mPrep; }
  const DecodingParams& getParams() const { return mParams; }
  // pairs queued since the last flush, full batches included
  size_t getQueuedPairs() const { return mPairs.size(); }
  // HMM::getBatchBuffer (HMM.hpp:215): the observations of the OPEN batch -- empty again whenever a batch fills up
  // (TESTS/test_HMM.cpp:49-79)
  std::vector<PairObservations> getBatchBuffer() const;

  // Every setStore* / setWrite* setter ends in updateOutputStructures() -> resetDecoding() as in the reference
  // (HMM.cpp:1733-1757, 1759-1800): the posterior sums are zeroed and the per-pair files reopened (truncated).  Full
  // batches the reference would have decoded by then are decoded first (flush).
  void setStorePerPairPosteriorMean(bool v);
  void setStorePerPairMap(bool v);
  void setStorePerPairPosterior(bool v);
  void setStoreSumOfPosterior(bool v);
  // HMM.hpp:287,293: per-pair posterior means / MAP states of every decoded pair as text, one row per pair, to
  // <outFileRoot>.perPairPosteriorMeans.gz / .perPairMAP.gz (ASMC mode; opened by resetDecoding, HMM.cpp:259-271,
  // written batch by batch, HMM.cpp:1412-1420, closed by finishDecoding, HMM.cpp:515-524)
  const std::vector<float>& getExpectedCoalTimes() const { return mExpectedCoalTimes; }
  void setWritePerPairPosteriorMean(bool v = true);
  void setWritePerPairMap(bool v = true);
  void resetDecoding(); // HMM.cpp:258-280
  void updateOutputStructures(); // HMM.cpp:1733-1757

  // Multi-GPU: this process decodes shard `rank` of `world` -- a contiguous range of the job's batches (whole
  // batches, so batch windows are those of a single-device run).  Output goes to "<file>.part<rank>of<world>";
  // parts concatenated in rank order are byte-for-byte a valid gzip stream of the single-device content
  // (only rank 0 writes the binary header).  Call before decodeAll.
  void setShard(int rank, int world);
  int shardRank() const { return mShardRank; }
  int shardWorld() const { return mShardWorld; }
  // [first, last) batch ordinals of this shard out of nBatches
  std::pair<unsigned long long, unsigned long long> shardBatchRange(unsigned long long nBatches) const;
  std::string ibdFileName(int jobs, int jobInd) const;
  int batchSize() const { return mBatchSize; }
  // the device context of this HMM (opened on first use; the identification step runs on it before any decode)
  fsmc_ctx* engine();
  // tell the engine how many pair-sites the coming flushes will decode (fsmc_ctx_expect_work: workspace policy)
  void announceWork(double pairSites);

  // keep emitted IBD records in memory as well (tests, benchmarks, the multi-GPU gather)
  void setKeepIbdRecords(bool v) { mKeepRecords = v; }
  // false: decodeAll / the hashing driver open no IBD file -- the records are only kept (setKeepIbdRecords) and leave
  // the process through the record gather (multi-GPU runs whose ranks send their records to rank 0 over RCCL)
  void setWriteIbdFile(bool v) { mWriteIbdFile = v; }
  // the file a single-device run of this job would have written, from records gathered in output order (rank 0 of a
  // multi-GPU run): header (binary output) and every record through the same formatter as writeIbd
  void writeIbdRecordsTo(const std::string& fileName, const std::vector<fsmc_pair>& pairs,
                         const std::vector<fsmc_ibd_record>& records) const;
  const std::vector<fsmc_ibd_record>& getIbdRecords() const { return mKeptRecords; }
  const std::vector<fsmc_pair>& getIbdRecordPairs() const { return mKeptPairs; }
  // ordinal of every kept record's pair among all the pairs this HMM has decoded (the candidate number in hashing mode)
  const std::vector<uint64_t>& getIbdRecordOrdinals() const { return mKeptOrdinals; }
  unsigned long long getNumSegmentsDetected() const { return mSegmentsDetected; }

  // text of one IBD record, exactly as HMM::writePairIBD formats it (HMM.cpp:1116-1144)
  std::string formatIbdRecord(const fsmc_pair& pr, const fsmc_ibd_record& r) const;
  // ... and of n records in order (record i belongs to pairs[pairOf ? pairOf[i] : i]): the same text, formatted by several
  // threads, each through one stream (a stream per record is most of what 20 000 records cost: 50 of 73 ms on the C2 job)
  std::string formatIbdRecords(const fsmc_pair* pairs, const fsmc_ibd_record* records, size_t n,
                               const uint32_t* pairOf = nullptr) const;

private:
  void putIbdRecord(std::ostream& record, const fsmc_pair& pr, const fsmc_ibd_record& r) const;
  void init(int scalingSkip);
  void prepareEmissions(); // HMM.cpp:159-256
  void prepareModel();
  void ensureEngine();
  template <typename Fn> void forEachPairOfJob(int jobs, int jobInd, bool shardOnly, Fn&& fn) const;
  // [lo, hi) ordinals of the job's (or, shardOnly, this device's) pairs in the enumeration of HMM.cpp:325-357
  std::pair<unsigned long long, unsigned long long> pairRangeOfJob(int jobs, int jobInd, bool shardOnly) const;
  void queuePair(unsigned hapRowA, unsigned hapRowB);
  void closeBatch(bool last);
  void flush();
  void writeIbd(const fsmc_pair& pr, const fsmc_ibd_record& r);
  void emitIbd(gzFile file, const fsmc_pair& pr, const fsmc_ibd_record& r) const;
  void writeBinaryHeader(gzFile file) const; // HMM.cpp:383-401
  void openIbdFile(int jobs, int jobInd);

  Data mData;
  DecodingQuantities mDq;
  DecodingParams mParams;
  PreparedModel mPrep;
  int mBatchSize = 64;
  std::vector<bool> mUseCSFS;

  // engine
  fsmc_ctx* mCtx = nullptr;
  fsmc_model* mModel = nullptr;
  bool mHapsUploaded = false;

  // work list under construction
  std::vector<fsmc_pair> mPairs;
  std::vector<fsmc_group> mGroups;
  std::vector<uint32_t> mBatchFirstGroup; // posterior sums: first group of every closed batch (a batch > 64 pairs is several groups)
  size_t mBatchBegin = 0; // first pair of the open batch
  std::vector<unsigned> mFromBatch, mToBatch; // per slot of the open batch (hashing mode), HMM.cpp:491-493
  unsigned long mHashingCount = 0;            // "cpt"
  size_t mFlushThreshold = 1u << 20;
  // FSMC_HOST_TIMING=1: seconds spent in the phases of flush(), printed by finishFromHashing / finishDecoding
  double mTimeUpload = 0, mTimeDecode = 0, mTimeWrite = 0;
  int mShardRank = 0, mShardWorld = 1;

  // outputs
  gzFile mIbdFile = nullptr;
  int mIbdFd = -1; // its descriptor (whole gzip members of a flush's text are written through it: putIbdText)
  unsigned long long mSegmentsDetected = 0;
  DecodingReturnValues mReturn;
  DecodePairsReturnStruct mPairsReturn;
  bool mStoreMean = false, mStoreMap = false, mStorePosterior = false, mStoreSumOfPosterior = false;
  bool mWriteMean = false, mWriteMap = false;
  gzFile mMeanFile = nullptr, mMapFile = nullptr;
  int mMeanFd = -1, mMapFd = -1; // their descriptors (blocks of rows go out as gzip members of their own)
  uint64_t mPerPairRows = 0; // rows written to the per-pair files since they were opened (batch boundaries)
  void closePerPairFiles();
  bool mKeepRecords = false;
  bool mWriteIbdFile = true;
  std::vector<fsmc_ibd_record> mKeptRecords;
  std::vector<fsmc_pair> mKeptPairs;
  std::vector<uint64_t> mKeptOrdinals;
  uint64_t mPairsFlushed = 0; // pairs of the flushes before the current one
  std::vector<float> mExpectedCoalTimes;
};

// starts the HIP runtime's initialisation on a helper thread (drivers: before the input files are read); HMM::engine() waits
void warmUpDevice(int device);

// HMM.cpp:43-61: second column of an intervals file ("intervalStart expectedCoalescentTime intervalEnd" per line)
std::vector<float> readExpectedTimesFromIntervalsFile(const std::string& fileName);
bool isRegularFile(const std::string& path);

// helpers shared with the drivers (HmmUtils.cpp)
float roundMorgans(float value, int precision, float min);                      // HmmUtils.cpp:65-79
int roundPhysical(int value, int precision);                                    // HmmUtils.cpp:81-94
unsigned getFromPosition(const std::vector<float>& gen, unsigned from, float cmDist = 0.5f); // :153-164
unsigned getToPosition(const std::vector<float>& gen, unsigned to, float cmDist = 0.5f);     // :166-177
std::pair<unsigned long, unsigned long> hapToDipId(unsigned long hapId);        // :179-182
unsigned long dipToHapId(unsigned long ind, unsigned long hap);                 // :184-188
std::string indPlusHapToCombinedId(const std::string& indId, unsigned long hap);          // :190-198
std::pair<std::string, unsigned long> combinedIdToIndPlusHap(const std::string& combinedId); // :200-208
unsigned long getIndIdxFromIdString(const std::vector<std::string>& ids, const std::string& id); // :210-217

} // namespace fsmc_host
