// hashing.hpp -- the GERMLINE-style identification pre-filter of FastSMC (next-row f1 of the scope table):
// haplotypes whose 64-site words collide are candidate IBD pairs; collisions in consecutive words (with up to
// `gap` missing words) are merged into a match interval; matches of at least min_m centimorgans are handed to
// HMM::decodeFromHashing.  Reference: FastSMC.cpp:118-235 (word streaming), HASHING/SeedHash.hpp:29-136,
// ExtendHash.hpp:26-128, Match.hpp:29-83, Utils.cpp:22-34.
//
// Differences, by design: words are read from the packed genotype matrix already in memory (word w of haplotype h
// IS bits[h][w] -- equality of words is invariant under the per-site folding); candidates are emitted in a
// defined order -- ascending pair key (lower haplotype * n + higher haplotype) within each flush -- whereas the
// reference emits in boost::unordered_map iteration order, which is an internal detail of one Boost version
// (SURVEY.md fact 9).  Batch composition, hence the padded decode windows, follow from that order.
#pragma once

#include <cstdint>
#include <unordered_map>
#include <vector>

#include "data.hpp"
#include "decoding_params.hpp"
#include "hmm.hpp"

namespace fsmc_host
{

// asmc::cmBetween (HASHING/Utils.cpp:22-34)
double cmBetween(int w1, int w2, const std::vector<float>& geneticPositions, int wordSize);

// Match (HASHING/Match.hpp:29-83)
class Match
{
public:
  explicit Match(unsigned long wordSize = 64, int i = 0) : mStart(i), mEnd(i), mWordSize(wordSize) {}
  void extend(int w) { mEnd = w > mEnd ? w : mEnd; }
  void addGap() { mGaps++; }
  int start() const { return mStart; }
  int end() const { return mEnd; }
  void setStart(int w) { mStart = w; }
  void setEnd(int w) { mEnd = w; }
  unsigned getGaps() const { return mGaps; }
  unsigned long getWordSize() const { return mWordSize; }

private:
  int mStart, mEnd;
  unsigned long mWordSize;
  unsigned mGaps = 0;
};

struct HashingCandidate {
  unsigned hapA, hapB; // local haplotype rows (hapA < hapB), as passed to decodeFromHashing
  unsigned from, to;   // first site of the first word, last site of the last word
};

class HashingPrefilter
{
public:
  HashingPrefilter(const Data& data, const DecodingParams& params);
  // The candidates in emission order, from the GPU (fsmc_identify: every pair of the job is a lane's state machine
  // over its word equalities).  This is what FastSMC::run and the sharded runs use.
  std::vector<HashingCandidate> runOnDevice(fsmc_ctx* ctx) const;
  // The same stream from the host restatement of the reference's two hash maps (one word at a time): the checker of
  // the GPU step in tests/ -- no product path calls it.
  template <typename Sink> void run(Sink&& sink);
  unsigned long numWords() const { return mNumWords; }
  // the words the identification step compares, [haplotype row][word]: bit b of word w = the allele of the b-th site
  // of the word -- Individuals::setMarker(w, snp_ctr) / getWordHash (Individuals.hpp:46-59, FastSMC.cpp:176-186)
  const std::vector<uint64_t>& words() const { return mWords; }
  size_t numHaps() const { return mNumHaps; }

private:
  bool pairInJob(unsigned hapI, unsigned hapJ) const; // SeedHash.hpp:93-128 (hapJ < hapI, local rows)
  void flush(int priorTo, int currentWord, bool all, std::vector<HashingCandidate>& out);
  void extendSeeds(const std::vector<unsigned>& members, unsigned long w, int cur, unsigned long wordsRead);

  const Data& mData;
  const DecodingParams& mParams;
  std::vector<uint64_t> mWords; // [hap][word] (possibly MAF filtered)
  unsigned long mNumWords = 0;
  unsigned mWordSize = 64;
  size_t mNumHaps = 0;
  std::unordered_map<uint64_t, Match> mExtend; // key = lower * n + higher
};

void runHashing(const Data& data, const DecodingParams& params, HMM& hmm);
// the candidate list alone, in emission order: from the GPU (callers that want to shard it across devices) ...
std::vector<HashingCandidate> hashingCandidatesDevice(const Data& data, const DecodingParams& params, int device);
// ... and from the host restatement (tests only)
std::vector<HashingCandidate> hashingCandidates(const Data& data, const DecodingParams& params);

} // namespace fsmc_host
