// fsmc_identify.h -- the identification step of FastSMC on the GPU (scope row f1): which haplotype pairs share
// 64-site words over a long enough stretch to be worth decoding.  Reference: FastSMC.cpp:118-235 (word streaming),
// HASHING/SeedHash.hpp:29-136 (haplotypes with equal words form a seed; the pairs of a seed that belong to the job),
// ExtendHash.hpp:26-128 (a pair's matching words are merged into one interval while no more than `gap` words in a row
// are missing; a word with too few distinct values extends every open interval instead), Match.hpp:29-83 and
// Utils.cpp:22-34 (an interval is reported when it spans at least min_m centimorgans).
//
// The reference walks the words one at a time with two hash maps; what it computes per pair is a tiny state machine
// over that pair's sequence of word equalities, and pairs are independent.  On the GPU every pair of the job is a
// lane's state machine: a workgroup takes a 32 x 32 tile of haplotype pairs, streams both sides' words through LDS
// 32 words at a time and each thread walks four pairs.  Integer compares only; the word matrix ([hap][word], a few
// MB) is read once per tile row/column and lives in L2 -- the work is n^2/2 * words compares, LDS-read bound.
// A chunk costs a pair three VALU instructions per word (compare, select, or into a 32-bit equality mask); the state
// machine itself only runs for the pairs that share a word in the chunk or have an interval open.
//   pass 1  id_dup_kernel         bit (w, j) = some haplotype i < j has the same word w            (all pairs)
//   pass 2  id_complexity_kernel  word w is used iff  distinct(w) / n > skip   (SeedHash size / individuals,
//                                 FastSMC.cpp:208-219), distinct(w) = n - popcount(bits of w)
//   pass 3  id_match_kernel       the state machines; reported intervals go to a record buffer (atomic append) with the
//                                 word at which the reference would have flushed them -- the host orders the records by
//                                 (flush word, lower * n + higher), the emission order this product defines
//                                 (fastsmc_amd/csrc/host/hashing.hpp)
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/fastsmc_hip.h"

namespace fsmc
{

constexpr int kIdTile = 32;  // haplotypes per tile side
constexpr int kIdChunk = 32; // words per LDS chunk = bits of an equality mask
constexpr int kIdThreads = 256;

struct IdParams {
  const unsigned long long* words; // [nHaps][nWords]
  unsigned nHaps, nWords;
  const unsigned* globalId; // [nHaps] haplotype number in the whole file (Individual::getIdNum per haplotype)
  fsmc_job_window job;
  const float* gen; // [nSites] genetic positions (Morgans)
  unsigned nSites;
  int gap;
  float skip, minM;
  unsigned* dupBits;   // [chunks][hapStride]: bit w%32 of entry (w/32, j) = some haplotype i < j has the same word w
  unsigned hapStride;  // nHaps rounded up to the tile
  unsigned* usedBits;  // [chunks]: bit w%32 of entry w/32 = word w takes part (enough distinct values)
  fsmc_candidate* out;
  unsigned cap;
  unsigned* count;
  unsigned wordSize;          // sites per word (DecodingParams::hashingWordSize)
  const unsigned char* depth; // max_seeds != 0: [word][hapStride] depth at which a haplotype's seed of that word stops
                              // being split (fsmc_identify_seeds.hip); else null
};

// SeedHash.hpp:93-128 with ind_i = the later and ind_j = the earlier haplotype of the pair
__device__ __forceinline__ bool idPairInJob(const fsmc_job_window& jw, const unsigned idI, const unsigned idJ)
{
  const unsigned bi = (jw.w_i - 1u) * jw.window_size, bj = (jw.w_j - 1u) * jw.window_size;
  if (jw.last_job) {
    return idI >= bi && idJ >= bj && idJ < bj + (idI - bi);
  }
  if (idI >= bi && idI < bi + jw.window_size && idJ >= bj && idJ < bj + jw.window_size) {
    const bool below = idJ < bj + (idI - bi);
    return jw.j_above_diag ? below : !below;
  }
  return false;
}

// one 32-haplotype x 32-word block of the word matrix into LDS, transposed to [word][hap] (+1: the column reads of a
// wave then fall on different banks); zero beyond the matrix
__device__ __forceinline__ void idLoadTile(unsigned long long (&dst)[kIdChunk][kIdTile + 1], const IdParams& p,
                                           const unsigned hap0, const unsigned word0)
{
#pragma unroll
  for (int q = 0; q < (kIdTile * kIdChunk) / kIdThreads; ++q) {
    const unsigned idx = threadIdx.x + q * kIdThreads;
    const unsigned h = idx / kIdChunk, w = idx % kIdChunk;
    unsigned long long v = 0;
    if (hap0 + h < p.nHaps && word0 + w < p.nWords) {
      v = p.words[(size_t)(hap0 + h) * p.nWords + word0 + w];
    }
    dst[w][h] = v;
  }
}

// Equality masks of this thread's four pairs over the 32 words of the chunk in LDS: bit w of m[r] = the words w of
// haplotypes (ty + 8r) of tile A and tx of tile B are equal.  Five LDS reads serve four pairs; three VALU
// instructions per pair and word -- the whole cost of a chunk for the (many) pairs that share no word in it.
__device__ __forceinline__ void idMasks(const unsigned long long (&As)[kIdChunk][kIdTile + 1],
                                        const unsigned long long (&Bs)[kIdChunk][kIdTile + 1], const unsigned tx,
                                        const unsigned ty, unsigned (&m)[4])
{
  m[0] = m[1] = m[2] = m[3] = 0u;
#pragma unroll 8
  for (int w = 0; w < kIdChunk; ++w) {
    const unsigned long long b = Bs[w][tx];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      m[r] |= (As[w][ty + 8u * r] == b ? 1u : 0u) << w;
    }
  }
}

// pass 1: for every word w and haplotype j, is there an i < j with the same word?  Grid: (tiles, tiles), upper triangle.
__global__ __launch_bounds__(kIdThreads, 4) void id_dup_kernel(const IdParams p)
{
  const unsigned bi = blockIdx.y, bj = blockIdx.x;
  if (bi > bj) {
    return;
  }
  __shared__ unsigned long long As[kIdChunk][kIdTile + 1];
  __shared__ unsigned long long Bs[kIdChunk][kIdTile + 1];
  const unsigned tx = threadIdx.x & 31u, ty = threadIdx.x >> 5;
  const unsigned j = bj * kIdTile + tx;
  unsigned ok[4]; // all ones / zero
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const unsigned i = bi * kIdTile + ty + 8u * r;
    ok[r] = (i < j && j < p.nHaps) ? ~0u : 0u;
  }
  for (unsigned w0 = 0, chunk = 0; w0 < p.nWords; w0 += kIdChunk, ++chunk) {
    __syncthreads();
    idLoadTile(As, p, bi * kIdTile, w0);
    idLoadTile(Bs, p, bj * kIdTile, w0);
    __syncthreads();
    unsigned m[4];
    idMasks(As, Bs, tx, ty, m);
    unsigned any = (m[0] & ok[0]) | (m[1] & ok[1]) | (m[2] & ok[2]) | (m[3] & ok[3]);
    const unsigned nw = p.nWords - w0;
    if (nw < (unsigned)kIdChunk) {
      any &= (1u << nw) - 1u; // (beyond the last word both tiles hold zeros)
    }
    if (any != 0u) {
      atomicOr(&p.dupBits[(size_t)chunk * p.hapStride + j], any);
    }
  }
}

// pass 2: SeedHash::size() / individuals > skip (FastSMC.cpp:208-212), both as float.  One workgroup per chunk of 32
// words: distinct(w) = nHaps - #{j: bit w of dupBits[chunk][j]}.
__global__ __launch_bounds__(kIdThreads) void id_complexity_kernel(const IdParams p)
{
  __shared__ unsigned dups[kIdChunk];
  const unsigned chunk = blockIdx.x;
  if (threadIdx.x < (unsigned)kIdChunk) {
    dups[threadIdx.x] = 0u;
  }
  __syncthreads();
  unsigned cnt[kIdChunk];
#pragma unroll
  for (int b = 0; b < kIdChunk; ++b) {
    cnt[b] = 0u;
  }
  for (unsigned j = threadIdx.x; j < p.nHaps; j += kIdThreads) {
    const unsigned v = p.dupBits[(size_t)chunk * p.hapStride + j];
#pragma unroll
    for (int b = 0; b < kIdChunk; ++b) {
      cnt[b] += (v >> b) & 1u;
    }
  }
#pragma unroll
  for (int b = 0; b < kIdChunk; ++b) {
    if (cnt[b]) {
      atomicAdd(&dups[b], cnt[b]);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned used = 0u;
    for (unsigned b = 0; b < (unsigned)kIdChunk && chunk * kIdChunk + b < p.nWords; ++b) {
      const unsigned seeds = p.nHaps - dups[b];
      if ((float)seeds / (float)p.nHaps > p.skip) {
        used |= 1u << b;
      }
    }
    p.usedBits[chunk] = used;
  }
}

// pass 3: the per-pair state machines.
// Reports are staged in LDS and appended to the record buffer a block at a time (one global atomic per flush instead
// of one per record: ten million appends to one counter were 95 % of this kernel's time).
constexpr int kIdStage = 512;

// One interval has run out: Match::print (Match.hpp:42-52) with cmBetween (Utils.cpp:22-34).
struct IdSink {
  const float* gen;
  unsigned nSites;
  unsigned wordSize;
  float minM;
  fsmc_candidate* out;
  unsigned cap;
  unsigned* count;
  fsmc_candidate* stage; // LDS
  unsigned* nStaged;     // LDS
};
__device__ __forceinline__ void idReport(const IdSink& p, const unsigned hapA, const unsigned hapB, const int start,
                                      const int end, const int flushWord)
{
  fsmc_candidate* const stage = p.stage;
  unsigned* const nStaged = p.nStaged;
  const size_t s0 = (size_t)p.wordSize * (size_t)start;
  size_t s1 = (size_t)p.wordSize * (size_t)end + (p.wordSize - 1u);
  if (s1 > (size_t)p.nSites - 1) {
    s1 = (size_t)p.nSites - 1;
  }
  const double len = 100.0 * (double)(p.gen[s1] - p.gen[s0]);
  if (len >= (double)p.minM) {
    fsmc_candidate c;
    c.hap_a = hapA;
    c.hap_b = hapB;
    c.from = (unsigned)start * p.wordSize;
    c.to = (unsigned)end * p.wordSize + (p.wordSize - 1u);
    c.flush_word = (unsigned)flushWord;
    const unsigned slot = atomicAdd(nStaged, 1u);
    if (slot < (unsigned)kIdStage) {
      stage[slot] = c;
    } else { // the stage is full: straight to the record buffer
      const unsigned idx = atomicAdd(p.count, 1u);
      if (idx < p.cap) {
        p.out[idx] = c;
      }
    }
  }
}

__global__ __launch_bounds__(kIdThreads, 4) void id_match_kernel(const IdParams p)
{
  const unsigned bi = blockIdx.y, bj = blockIdx.x;
  if (bi > bj) {
    return;
  }
  __shared__ unsigned long long As[kIdChunk][kIdTile + 1];
  __shared__ unsigned long long Bs[kIdChunk][kIdTile + 1];
  __shared__ fsmc_candidate stage[kIdStage];
  __shared__ unsigned nStaged, stageBase;
  __shared__ int anyInJob;
  const unsigned tx = threadIdx.x & 31u, ty = threadIdx.x >> 5;
  const unsigned j = bj * kIdTile + tx;
  unsigned ok[4]; // all ones / zero
  int start[4], end[4];
  bool open[4];
  if (threadIdx.x == 0) {
    anyInJob = 0;
    nStaged = 0u;
  }
  __syncthreads();
  bool mine = false;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const unsigned i = bi * kIdTile + ty + 8u * r;
    const bool in = i < j && j < p.nHaps && idPairInJob(p.job, p.globalId[j], p.globalId[i]);
    ok[r] = in ? ~0u : 0u;
    mine = mine || in;
    open[r] = false;
    start[r] = 0;
    end[r] = 0;
  }
  if (mine) {
    anyInJob = 1;
  }
  __syncthreads();
  if (!anyInJob) {
    return; // no pair of this tile belongs to the job
  }
  const IdSink sink = {p.gen, p.nSites, p.wordSize, p.minM, p.out, p.cap, p.count, stage, &nStaged};
  auto report = [&](const int r, const int flushWord) {
    idReport(sink, bi * kIdTile + ty + 8u * r, j, start[r], end[r], flushWord);
  };
  // all threads, between two barriers: move the staged records to the record buffer
  auto flushStage = [&]() {
    const unsigned n = nStaged < (unsigned)kIdStage ? nStaged : (unsigned)kIdStage;
    __syncthreads();
    if (threadIdx.x == 0) {
      stageBase = atomicAdd(p.count, n);
      nStaged = 0u;
    }
    __syncthreads();
    for (unsigned i = threadIdx.x; i < n; i += kIdThreads) {
      if (stageBase + i < p.cap) {
        p.out[stageBase + i] = stage[i];
      }
    }
  };
  for (unsigned w0 = 0, chunk = 0; w0 < p.nWords; w0 += kIdChunk, ++chunk) {
    __syncthreads();
    if (nStaged >= (unsigned)kIdStage / 2) { // (uniform: read behind the barrier, written only before it)
      flushStage();
      __syncthreads();
    }
    idLoadTile(As, p, bi * kIdTile, w0);
    idLoadTile(Bs, p, bj * kIdTile, w0);
    __syncthreads();
    unsigned m[4];
    idMasks(As, Bs, tx, ty, m);
    const unsigned used = p.usedBits[chunk]; // (zero beyond the last word)
    const unsigned nw = p.nWords - w0 < (unsigned)kIdChunk ? p.nWords - w0 : (unsigned)kIdChunk;
    const unsigned valid = nw < (unsigned)kIdChunk ? (1u << nw) - 1u : ~0u;
    const int last = (int)(w0 + nw) - 1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      unsigned eq = m[r] & ok[r] & used;
      if (eq == 0u && !open[r]) {
        continue; // nothing opens and nothing is open: the state of this pair does not change in this chunk
      }
      if (used == valid) {
        // Every word of the chunk takes part: the state only changes at the pair's matching words and where its
        // open interval runs out, `gap` + 1 words after its end (clearPairsPriorTo(cur - gap) is called at every word
        // and reports an interval as soon as end < cur - gap, ExtendHash.hpp:85-105) -- walk those events instead of
        // the words.  (On entry an open interval has not run out before this chunk: end + gap + 1 >= w0.)
        // (a run of consecutive matching words is one event: inside it nothing can run out; the end of the chunk is
        //  the last event)
        bool more = true;
        while (more) {
          int cur = last + 1, runEnd = 0;
          const bool isRun = eq != 0u;
          if (isRun) {
            const int lo = __ffs((int)eq) - 1;
            const unsigned run = eq >> lo; // bit 0 set
            const int len = run == ~0u ? 32 : __ffs((int)~run) - 1;
            eq = len + lo >= 32 ? 0u : eq & (~0u << (lo + len));
            cur = (int)w0 + lo;
            runEnd = cur + len - 1;
          } else {
            more = false;
          }
          if (open[r] && cur > end[r] + p.gap + 1) {
            report(r, end[r] + p.gap + 1);
            open[r] = false;
          }
          if (isRun) {
            if (!open[r]) {
              open[r] = true;
              start[r] = cur;
            }
            end[r] = runEnd;
          }
        }
      } else {
        for (unsigned w = 0; w < nw; ++w) {
          const int cur = (int)(w0 + w);
          if ((used >> w) & 1u) {
            // ExtendHash::extendPair (ExtendHash.hpp:73-80), then clearPairsPriorTo(cur - gap) (85-105)
            if ((eq >> w) & 1u) {
              if (!open[r]) {
                open[r] = true;
                start[r] = cur;
              }
              end[r] = cur;
            }
            if (open[r] && end[r] < cur - p.gap) {
              report(r, cur);
              open[r] = false;
            }
          } else if (open[r]) {
            // a word with too few distinct values: every open interval is carried over it (ExtendHash.hpp:100-104)
            end[r] = cur;
          }
        }
      }
    }
  }
  // clearAllPairs (ExtendHash.hpp:108-116): what is still open is reported after the last word
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (open[r]) {
      report(r, (int)p.nWords);
    }
  }
  __syncthreads();
  if (nStaged != 0u) {
    flushStage();
  }
}

// pass 3, the general form: matches keyed by individual pairs (haploid = false, ExtendHash.hpp:47-70) and/or seeds
// split by the words ahead (max_seeds, SeedHash.hpp:41-85).  The same tiles; what differs from id_match_kernel:
//   * a thread's four haplotype pairs are (row[r], col[r]) and feed NS state machines: haploid -- the four pairs of
//     id_match_kernel, one machine each; not haploid -- the 2 x 2 haplotype pairs of ONE pair of individuals (a tile is
//     16 x 16 individuals), one machine: any of the four extends it, the candidate names the first haplotype of each
//     (locationToPair), and on the diagonal the two haplotypes of one individual are a pair;
//   * with max_seeds a pair is extended at word c only if it also shares the D words after it, D = depth[c][row], and
//     then to word c + D: the equality masks of the NEXT chunk are computed one chunk ahead (D <= read_ahead - 1 <= 31);
//   * the words are walked one by one (an interval's end can lie ahead of the current word).
template <bool HAPLOID>
__global__ __launch_bounds__(kIdThreads, 4) void id_match_general_kernel(const IdParams p)
{
  const unsigned bi = blockIdx.y, bj = blockIdx.x;
  if (bi > bj) {
    return;
  }
  __shared__ unsigned long long As[kIdChunk][kIdTile + 1];
  __shared__ unsigned long long Bs[kIdChunk][kIdTile + 1];
  __shared__ unsigned char depthA[kIdChunk][kIdTile];
  __shared__ fsmc_candidate stage[kIdStage];
  __shared__ unsigned nStaged, stageBase;
  __shared__ int anyInJob;
  constexpr int NS = HAPLOID ? 4 : 1;
  unsigned row[4], col[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (HAPLOID) {
      row[r] = (threadIdx.x >> 5) + 8u * r;
      col[r] = threadIdx.x & 31u;
    } else {
      row[r] = 2u * (threadIdx.x >> 4) + (unsigned)(r >> 1);
      col[r] = 2u * (threadIdx.x & 15u) + (unsigned)(r & 1);
    }
  }
  unsigned ok[4];
  int start[NS], end[NS];
  bool open[NS];
  if (threadIdx.x == 0) {
    anyInJob = 0;
    nStaged = 0u;
  }
  __syncthreads();
  bool mine = false;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const unsigned i = bi * kIdTile + row[r], j = bj * kIdTile + col[r];
    const bool in = i < j && j < p.nHaps && idPairInJob(p.job, p.globalId[j], p.globalId[i]);
    ok[r] = in ? ~0u : 0u;
    mine = mine || in;
  }
#pragma unroll
  for (int m = 0; m < NS; ++m) {
    open[m] = false;
    start[m] = 0;
    end[m] = 0;
  }
  if (mine) {
    anyInJob = 1;
  }
  __syncthreads();
  if (!anyInJob) {
    return;
  }
  const IdSink sink = {p.gen, p.nSites, p.wordSize, p.minM, p.out, p.cap, p.count, stage, &nStaged};
  auto report = [&](const int m, const int flushWord) {
    const unsigned a = bi * kIdTile + (HAPLOID ? row[m] : row[0]); // (row[0], col[0]): haplotype 1 of each individual
    const unsigned b = bj * kIdTile + (HAPLOID ? col[m] : col[0]);
    idReport(sink, a, b, start[m], end[m], flushWord);
  };
  auto flushStage = [&]() {
    const unsigned n = nStaged < (unsigned)kIdStage ? nStaged : (unsigned)kIdStage;
    __syncthreads();
    if (threadIdx.x == 0) {
      stageBase = atomicAdd(p.count, n);
      nStaged = 0u;
    }
    __syncthreads();
    for (unsigned i = threadIdx.x; i < n; i += kIdThreads) {
      if (stageBase + i < p.cap) {
        p.out[stageBase + i] = stage[i];
      }
    }
  };
  auto masks = [&](unsigned (&m)[4]) {
    m[0] = m[1] = m[2] = m[3] = 0u;
#pragma unroll 8
    for (int w = 0; w < kIdChunk; ++w) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        m[r] |= (As[w][row[r]] == Bs[w][col[r]] ? 1u : 0u) << w;
      }
    }
  };
  unsigned mCur[4], mNext[4];
  idLoadTile(As, p, bi * kIdTile, 0u);
  idLoadTile(Bs, p, bj * kIdTile, 0u);
  __syncthreads();
  masks(mCur);
  for (unsigned w0 = 0, chunk = 0; w0 < p.nWords; w0 += kIdChunk, ++chunk) {
    __syncthreads();
    if (nStaged >= (unsigned)kIdStage / 2) {
      flushStage();
      __syncthreads();
    }
    // the next chunk's words (zeros beyond the matrix: never looked at, c + D < words read) and this chunk's depths
    idLoadTile(As, p, bi * kIdTile, w0 + kIdChunk);
    idLoadTile(Bs, p, bj * kIdTile, w0 + kIdChunk);
#pragma unroll
    for (int q = 0; q < (kIdTile * kIdChunk) / kIdThreads; ++q) {
      const unsigned idx = threadIdx.x + q * kIdThreads;
      const unsigned w = idx / kIdTile, h = idx % kIdTile;
      unsigned char d = 0;
      if (p.depth && w0 + w < p.nWords) {
        d = p.depth[(size_t)(w0 + w) * p.hapStride + bi * kIdTile + h];
      }
      depthA[w][h] = d;
    }
    __syncthreads();
    masks(mNext);
    const unsigned used = p.usedBits[chunk];
    const unsigned nw = p.nWords - w0 < (unsigned)kIdChunk ? p.nWords - w0 : (unsigned)kIdChunk;
    unsigned long long eq[4];
    unsigned long long any = 0ull;
    bool anyOpen = false;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      eq[r] = (unsigned long long)(mCur[r] & ok[r]) | ((unsigned long long)(mNext[r] & ok[r]) << 32);
      any |= eq[r] & (unsigned long long)used;
      mCur[r] = mNext[r];
    }
#pragma unroll
    for (int m = 0; m < NS; ++m) {
      anyOpen = anyOpen || open[m];
    }
    if (any == 0ull && !anyOpen) {
      continue; // (no barrier is skipped: the loop's barriers are above)
    }
    for (unsigned w = 0; w < nw; ++w) {
      const int cur = (int)(w0 + w);
      if ((used >> w) & 1u) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = HAPLOID ? r : 0;
          if ((eq[r] >> w) & 1ull) {
            // SeedHash::extendAllPairs / subHash: the pair must share the D words after this one as well
            const unsigned D = depthA[w][row[r]];
            const unsigned long long need = (2ull << D) - 1ull;
            if (((eq[r] >> w) & need) == need) {
              const int to = cur + (int)D;
              if (!open[m]) { // ExtendHash::extendPair (ExtendHash.hpp:73-80): a new interval starts at the CURRENT word
                open[m] = true;
                start[m] = cur;
                end[m] = to;
              } else if (to > end[m]) {
                end[m] = to;
              }
            }
          }
        }
#pragma unroll
        for (int m = 0; m < NS; ++m) {
          if (open[m] && end[m] < cur - p.gap) { // clearPairsPriorTo(cur - gap), ExtendHash.hpp:85-105
            report(m, cur);
            open[m] = false;
          }
        }
      } else {
#pragma unroll
        for (int m = 0; m < NS; ++m) {
          if (open[m]) {
            end[m] = cur; // extendAllPairsTo (ExtendHash.hpp:100-104) ASSIGNS the current word
          }
        }
      }
    }
  }
#pragma unroll
  for (int m = 0; m < NS; ++m) {
    if (open[m]) {
      report(m, (int)p.nWords);
    }
  }
  __syncthreads();
  if (nStaged != 0u) {
    flushStage();
  }
}

} // namespace fsmc
