#include "hashing.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

namespace fsmc_host
{

double cmBetween(const int w1, const int w2, const std::vector<float>& gen, const int wordSize)
{
  const std::size_t start = static_cast<std::size_t>(wordSize) * w1;
  const std::size_t end = std::min<std::size_t>(static_cast<std::size_t>(wordSize) * w2 + wordSize - 1, gen.size() - 1ul);
  return 100.0 * (gen[end] - gen[start]);
}

HashingPrefilter::HashingPrefilter(const Data& data, const DecodingParams& params) : mData(data), mParams(params)
{
  if (params.hashingWordSize < 1 || params.hashingWordSize > 64) {
    // Individuals::getWordHash is bitset::to_ulong() (Individuals.hpp:46-50): a word is one 64-bit integer
    throw std::runtime_error("hashingWordSize must be 1..64 (a word is hashed as one 64-bit integer)");
  }
  if (params.constReadAhead < 1 || (params.max_seeds > 0 && params.constReadAhead > 32)) {
    throw std::runtime_error("constReadAhead must be at least 1, and at most 32 when max_seeds is set");
  }
  mWordSize = static_cast<unsigned>(params.hashingWordSize);
  mNumHaps = data.numHapRows();
  const size_t S = static_cast<size_t>(data.sites);
  const size_t W = mWordSize;
  if (params.min_maf <= 0.f && W == 64) {
    // only complete words are hashed (FastSMC.cpp:186-195: a word counts once its 64th site has been read)
    mNumWords = S / 64;
    mWords.resize(mNumHaps * mNumWords);
    for (size_t h = 0; h < mNumHaps; ++h) {
      for (size_t w = 0; w < mNumWords; ++w) {
        mWords[h * mNumWords + w] = data.bits[h * data.wordsPerHap + w];
      }
    }
  } else {
    // sites failing the MAF filter are skipped by the word stream (FastSMC.cpp:154-172); positions derived from
    // word numbers then count kept sites only, exactly as in the reference
    std::vector<size_t> kept;
    for (size_t s = 0; s < S; ++s) {
      if (params.min_maf > 0.f) {
        // FastSMC.cpp:154-166: frequency of allele '1' over every haplotype of the file
        const int total = data.totalSamplesCount[s];
        const int ones = (data.foldToMinorAlleles && data.siteWasFlippedDuringFolding[s])
                             ? total - data.derivedAlleleCounts[s]
                             : data.derivedAlleleCounts[s];
        const auto maf = static_cast<float>(ones / static_cast<double>(total));
        if (maf < params.min_maf || maf > 1 - params.min_maf) {
          continue;
        }
      }
      kept.push_back(s);
    }
    // bit b of word w = the allele of the b-th kept site of the word (Individuals::setMarker(w, snp_ctr))
    mNumWords = kept.size() / W;
    mWords.assign(mNumHaps * mNumWords, 0ull);
    for (size_t h = 0; h < mNumHaps; ++h) {
      for (size_t i = 0; i < mNumWords * W; ++i) {
        if (data.genotype(h, kept[i])) {
          mWords[h * mNumWords + i / W] |= 1ull << (i % W);
        }
      }
    }
  }
}

bool HashingPrefilter::pairInJob(const unsigned hapI, const unsigned hapJ) const
{
  // SeedHash.hpp:93-128 with ind_i = the higher and ind_j = the lower haplotype; getIdNum() is the haplotype's
  // number in the whole file (2 * sample line + 0/1).
  const unsigned idI = mData.globalHapId(hapI);
  const unsigned idJ = mData.globalHapId(hapJ);
  const unsigned ws = static_cast<unsigned>(mData.windowSize);
  const unsigned wi = static_cast<unsigned>(mData.w_i), wj = static_cast<unsigned>(mData.w_j);
  if (mParams.jobInd == mParams.jobs) {
    return idI >= (wi - 1) * ws && idJ >= (wj - 1) * ws && idJ < (wj - 1) * ws + (idI - (wi - 1) * ws);
  }
  if ((idI >= (wi - 1) * ws && idI < wi * ws) && (idJ >= (wj - 1) * ws && idJ < wj * ws)) {
    const bool below = idJ < (wj - 1) * ws + (idI - (wi - 1) * ws);
    return mData.is_j_above_diag ? below : !below;
  }
  return false;
}

void HashingPrefilter::flush(const int priorTo, const int currentWord, const bool all,
                             std::vector<HashingCandidate>& out)
{
  // ExtendHash::clearPairsPriorTo / clearAllPairs (ExtendHash.hpp:85-116) + Match::print (Match.hpp:42-52)
  std::vector<uint64_t> done;
  for (auto& kv : mExtend) {
    Match& m = kv.second;
    if (all || m.end() < priorTo) {
      done.push_back(kv.first);
    } else if (m.end() < currentWord) {
      m.addGap();
    }
  }
  std::sort(done.begin(), done.end()); // the defined emission order
  for (const uint64_t key : done) {
    const Match& m = mExtend.at(key);
    const int W = static_cast<int>(mWordSize);
    const double mlen = cmBetween(m.start(), m.end(), mData.geneticPositions, W);
    if (mlen >= mParams.min_m) {
      HashingCandidate c;
      // ExtendHash::locationToPair (ExtendHash.hpp:47-53): not haploid -- the first haplotype of each individual
      const unsigned scale = mParams.haploid ? 1u : 2u;
      c.hapA = scale * static_cast<unsigned>(key / mNumHaps);
      c.hapB = scale * static_cast<unsigned>(key % mNumHaps);
      c.from = static_cast<unsigned>(m.start() * W);
      c.to = static_cast<unsigned>(m.end() * W + W - 1);
      out.push_back(c);
    }
    mExtend.erase(key);
  }
}

std::vector<HashingCandidate> HashingPrefilter::runOnDevice(fsmc_ctx* ctx) const
{
  std::vector<HashingCandidate> out;
  if (mNumHaps < 2 || mNumWords == 0) {
    return out;
  }
  std::vector<uint32_t> ids(mNumHaps);
  for (size_t h = 0; h < mNumHaps; ++h) {
    ids[h] = mData.globalHapId(static_cast<unsigned>(h));
  }
  fsmc_job_window jw{};
  jw.window_size = static_cast<uint32_t>(mData.windowSize);
  jw.w_i = static_cast<uint32_t>(mData.w_i);
  jw.w_j = static_cast<uint32_t>(mData.w_j);
  jw.last_job = mParams.jobInd == mParams.jobs ? 1 : 0;
  jw.j_above_diag = mData.is_j_above_diag ? 1 : 0;
  std::vector<fsmc_candidate> buf(std::max<size_t>(1024, 4 * mNumHaps));
  size_t n = 0;
  fsmc_identify_opts opts{};
  opts.word_size = mWordSize;
  opts.haploid = mParams.haploid ? 1u : 0u;
  opts.max_seeds = mParams.max_seeds;
  opts.read_ahead = static_cast<uint32_t>(mParams.constReadAhead);
  auto call = [&]() {
    return fsmc_identify_ex(ctx, mWords.data(), static_cast<uint32_t>(mNumHaps), static_cast<uint32_t>(mNumWords),
                            ids.data(), &jw, mData.geneticPositions.data(),
                            static_cast<uint32_t>(mData.geneticPositions.size()), mParams.gap, mParams.skip,
                            mParams.min_m, &opts, buf.data(), buf.size(), &n);
  };
  int rc = call();
  if (rc == FSMC_EOVERFLOW) {
    // (nobody knows the count beforehand: the library finished the list on the device and kept it)
    buf.resize(n);
    rc = fsmc_identify_fetch(ctx, buf.data(), buf.size(), &n);
    if (rc != FSMC_OK) {
      rc = call();
    }
  }
  if (rc != FSMC_OK) {
    throw std::runtime_error(std::string("fsmc_identify: ") + fsmc_last_error(ctx));
  }
  out.resize(n);
  for (size_t i = 0; i < n; ++i) {
    out[i] = HashingCandidate{buf[i].hap_a, buf[i].hap_b, buf[i].from, buf[i].to};
  }
  return out;
}

void HashingPrefilter::extendSeeds(const std::vector<unsigned>& members, const unsigned long w, const int cur,
                                   const unsigned long wordsRead)
{
  // SeedHash::extendAllPairs (SeedHash.hpp:62-135) for the seeds of word w among `members`
  std::vector<std::pair<uint64_t, unsigned>> order(members.size());
  for (size_t i = 0; i < members.size(); ++i) {
    order[i] = {mWords[members[i] * mNumWords + w], members[i]};
  }
  std::sort(order.begin(), order.end());
  std::vector<unsigned> seed;
  for (size_t a = 0; a < order.size();) {
    size_t b = a;
    while (b < order.size() && order[b].first == order[a].first) {
      ++b;
    }
    if (mParams.max_seeds != 0 && b - a > static_cast<unsigned long>(mParams.max_seeds) && w + 1 < wordsRead) {
      // a large seed is split by the next word, only the pairs of the sub-seeds go on (SeedHash.hpp:41-55, 75-85)
      seed.clear();
      for (size_t i = a; i < b; ++i) {
        seed.push_back(order[i].second);
      }
      const std::vector<unsigned> sub = seed; // (the recursion reuses `seed`)
      extendSeeds(sub, w + 1, cur, wordsRead);
    } else {
      for (size_t i = a; i < b; ++i) {
        for (size_t ii = i + 1; ii < b; ++ii) {
          const unsigned lo = order[i].second, hi = order[ii].second; // sorted: lo < hi
          if (pairInJob(hi, lo)) {
            // ExtendHash::extendPair (ExtendHash.hpp:73-80) with pairToLocation (60-70)
            const uint64_t x = mParams.haploid ? lo : lo / 2, y = mParams.haploid ? hi : hi / 2;
            auto it = mExtend.emplace(x * mNumHaps + y, Match(mWordSize, cur)).first;
            it->second.extend(static_cast<int>(w)); // (a new interval is [cur, 0] extended to w >= cur)
          }
        }
      }
    }
    a = b;
  }
}

template <typename Sink> void HashingPrefilter::run(Sink&& sink)
{
  std::vector<std::pair<uint64_t, unsigned>> order(mNumHaps);
  std::vector<unsigned> everyone(mNumHaps);
  for (size_t h = 0; h < mNumHaps; ++h) {
    everyone[h] = static_cast<unsigned>(h);
  }
  std::vector<HashingCandidate> out;
  for (unsigned long w = 0; w < mNumWords; ++w) {
    // SeedHash: haplotypes with the same word form a seed (SeedHash.hpp:29-33)
    for (size_t h = 0; h < mNumHaps; ++h) {
      order[h] = {mWords[h * mNumWords + w], static_cast<unsigned>(h)};
    }
    std::sort(order.begin(), order.end());
    size_t seeds = 0;
    for (size_t a = 0; a < mNumHaps;) {
      size_t b = a;
      while (b < mNumHaps && order[b].first == order[a].first) {
        ++b;
      }
      ++seeds;
      a = b;
    }
    const int cur = static_cast<int>(w);
    // FastSMC.cpp:186-195: one more word is read per word processed, constReadAhead of them before the first
    const unsigned long wordsRead = std::min<unsigned long>(mNumWords, w + static_cast<unsigned long>(mParams.constReadAhead));
    if (static_cast<float>(seeds) / static_cast<float>(mNumHaps) > mParams.skip) {
      extendSeeds(everyone, w, cur, wordsRead);
      out.clear();
      flush(cur - mParams.gap, cur, false, out);
      for (const auto& c : out) {
        sink(c);
      }
    } else {
      // low-complexity word: every open match is carried over it (ExtendHash.hpp:100-104)
      for (auto& kv : mExtend) {
        kv.second.setEnd(cur);
      }
    }
  }
  out.clear();
  flush(0, 0, true, out);
  for (const auto& c : out) {
    sink(c);
  }
}

void runHashing(const Data& data, const DecodingParams& params, HMM& hmm)
{
  using Clock = std::chrono::steady_clock;
  const bool timing = std::getenv("FSMC_HOST_TIMING") != nullptr;
  Clock::time_point t0 = Clock::now();
  HashingPrefilter pf(data, params);
  const double tWords = std::chrono::duration<double>(Clock::now() - t0).count();
  t0 = Clock::now();
  const std::vector<HashingCandidate> all = pf.runOnDevice(hmm.engine());
  if (timing) {
    std::fprintf(stderr, "[fsmc host] hashing words %.3f s, identification (engine + fsmc_identify) %.3f s, %zu candidates\n",
                 tWords, std::chrono::duration<double>(Clock::now() - t0).count(), all.size());
  }
  // sharded: every rank runs the identification step and decodes a contiguous range of the resulting batches, so
  // each batch has the composition -- hence the window -- of a single-device run.  The ranges have equal pair-site
  // WEIGHT, not equal batch counts (the windows differ in length): the reference's own range rule total*r/R
  // (HMM.cpp:319-321) applied to the running weight, cut at batch boundaries.
  size_t first = 0, last = all.size();
  if (hmm.shardWorld() > 1) {
    const size_t B = static_cast<size_t>(hmm.batchSize());
    const size_t nBatches = (all.size() + B - 1) / B;
    std::vector<unsigned long long> upTo(nBatches + 1, 0ull); // pair-sites of the batches before batch i
    for (size_t b = 0; b < nBatches; ++b) {
      unsigned lo = ~0u, hi = 0u;
      const size_t e = std::min(all.size(), (b + 1) * B);
      for (size_t i = b * B; i < e; ++i) {
        lo = std::min(lo, all[i].from);
        hi = std::max(hi, all[i].to);
      }
      upTo[b + 1] = upTo[b] + static_cast<unsigned long long>(e - b * B) * (hi - lo + 1);
    }
    const auto world = static_cast<unsigned long long>(hmm.shardWorld());
    auto cut = [&](unsigned long long r) { // first batch whose preceding weight reaches total * r / world
      const unsigned long long want = upTo[nBatches] / world * r + upTo[nBatches] % world * r / world;
      return static_cast<size_t>(std::lower_bound(upTo.begin(), upTo.end(), want) - upTo.begin());
    };
    const auto r = static_cast<unsigned long long>(hmm.shardRank());
    const size_t bLo = r == 0 ? 0 : std::min(cut(r), nBatches);
    const size_t bHi = r + 1 == world ? nBatches : std::min(cut(r + 1), nBatches);
    first = std::min(all.size(), bLo * B);
    last = std::min(all.size(), std::max(bLo, bHi) * B);
  }
  // the candidates' windows, padded like a batch's (roughly: every batch decodes the union of its members' windows):
  // the size of the job, announced before its first flush (fsmc_ctx_expect_work)
  double pairSites = 0;
  for (size_t i = first; i < last; ++i) {
    pairSites += static_cast<double>(all[i].to - all[i].from + 1);
  }
  if (pairSites > 0) {
    hmm.announceWork(pairSites);
  }
  for (size_t i = first; i < last; ++i) {
    hmm.decodeFromHashing(all[i].hapA, all[i].hapB, all[i].from, all[i].to);
  }
}

std::vector<HashingCandidate> hashingCandidatesDevice(const Data& data, const DecodingParams& params, const int device)
{
  HashingPrefilter pf(data, params);
  fsmc_ctx* ctx = nullptr;
  if (fsmc_ctx_create(device, nullptr, &ctx) != FSMC_OK) {
    throw std::runtime_error(std::string("cannot open the MI355X engine: ") + fsmc_last_error(nullptr));
  }
  try {
    std::vector<HashingCandidate> out = pf.runOnDevice(ctx);
    fsmc_ctx_destroy(ctx);
    return out;
  } catch (...) {
    fsmc_ctx_destroy(ctx);
    throw;
  }
}

std::vector<HashingCandidate> hashingCandidates(const Data& data, const DecodingParams& params)
{
  HashingPrefilter pf(data, params);
  std::vector<HashingCandidate> all;
  pf.run([&](const HashingCandidate& c) { all.push_back(c); });
  return all;
}

} // namespace fsmc_host
