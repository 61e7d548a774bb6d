// fsmc_kernels_any.h -- the decode for models of ANY number of states (K > 1024: beyond the lane-per-pair family of
// fsmc_kernels.h and the wave-group kernel of fsmc_kernels_w2.h).  The reference has no limit on K
// (DecodingQuantities.cpp:72-78); the kernels that hold a pair's K-vectors in registers stop at 1024 states (eight waves
// of 128, two of whose 256 registers' worth of vectors live in scratch memory already).  This one holds them in the wave's workspace instead: lane = pair as everywhere, a K-vector is a row of
// [K/4][64 lanes] float4 in HBM / L2, and a step walks its states with real loops -- the reference's loops
// (HMM.cpp:787-830 forward, 943-1016 backward, HmmUtils.cpp:102-151 scaling, HMM.cpp:669-692 combine), the same
// operations in the same order, so the results are the oracle's bit for bit.  It moves several rows per step where the
// other kernels move half a row: a correct path for rare models, not a fast one.
//
// Same work list, same queue, same chunking as decode_kernel: pass B walks beta from the window's last site down and
// keeps a checkpoint row at every chunk boundary; pass A rebuilds a chunk's beta rows from its checkpoint and sweeps
// alpha through it, combining and feeding the consumer (IBD scan with segment ages, posterior dump, per-pair mean /
// MAP, sums over pairs) site by site.  Array and sequence mode (SEQ: a half-step across the gap with the homozygous
// emission row, then the site step; one scaling for the two, HMM.cpp:760-770, 915-925).
#pragma once

#include "fsmc_kernels.h"

namespace fsmc
{

constexpr int kAnyExtraRows = 9;  // per wave, beside the chunk buffer and the checkpoints: alpha (2), beta (2),
                                  // a temporary, alpha*beta, per-state sums of open segments, the half-step (SEQ)

__device__ __forceinline__ float4 anyLd(const float4* row, const int k4, const int lane)
{
  return row[(size_t)k4 * kWave + lane];
}
__device__ __forceinline__ void anySt(float4* row, const int k4, const int lane, const float4 v)
{
  row[(size_t)k4 * kWave + lane] = v;
}
__device__ __forceinline__ float& anyAt(float4& v, const int i)
{
  return reinterpret_cast<float*>(&v)[i];
}
__device__ __forceinline__ float anyGet(const float4& v, const int i)
{
  return reinterpret_cast<const float*>(&v)[i];
}
// A row is K4 float4 of values and one more float4 whose first float is the row's SCALE: the vector the row stands for
// is value * scale.  The reference scales a vector right after it computes it (sum over the states, 1.0f / sum, multiply:
// HmmUtils.cpp:102-151); here the multiply happens when the row is read -- the same multiplication on the same bits -- and
// a pass over the row in memory is saved (a step is HBM-bound on its own temporaries).  Un-normalised rows (the half-steps
// of sequence mode) have scale 1.0f: x * 1.0f is exact.
__device__ __forceinline__ float anyScaleOf(const float4* row, const int K4, const int lane)
{
  return anyLd(row, K4, lane).x;
}
__device__ __forceinline__ void anySetScale(float4* row, const int K4, const int lane, const float sc)
{
  anySt(row, K4, lane, make_float4(sc, 0.f, 0.f, 0.f));
}
__device__ __forceinline__ void anyCopy(float4* dst, const float4* src, const int K4, const int lane)
{
  for (int k4 = 0; k4 <= K4; ++k4) {
    anySt(dst, k4, lane, anyLd(src, k4, lane));
  }
}

// HMM::getPreviousBetaBatched (HMM.cpp:943-1016) + the scaling sum: out = beta of the site before `last`'s, its values
// un-scaled and its scale 1 / (sum over the states, k ascending from 0.f) -- or 1 when the step is not normalised.
// e: this lane's emission row of last's site ([KP/4] float4); D, B, U, RR: the step's table rows; BU: temporary.
// Two passes: down (vec = last*e on the fly, BU kept) and up (vec again -- the same products --, BL, the row, its sum).
__device__ __forceinline__ void anyBetaStep(const int K, const float4* e, const float* D, const float* B, const float* U,
                                            const float* RR, const float4* last, float4* out, float4* BU,
                                            const bool normalise, const int lane)
{
  const int K4 = (K + 3) >> 2;
  const float scL = anyScaleOf(last, K4, lane);
  // BU[K-1] = 0, BU[k] = U[k]*vec[k+1] + RR[k]*BU[k+1], from the top down
  float buAbove = 0.f, vecAbove = 0.f;
  for (int k4 = K4 - 1; k4 >= 0; --k4) {
    const float4 l = anyLd(last, k4, lane), em = e[k4];
    float4 bu = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = 3; i >= 0; --i) {
      const int k = 4 * k4 + i;
      if (k < K) {
        float x = 0.f;
        if (k < K - 1) {
          x = U[k] * vecAbove + RR[k] * buAbove;
        }
        anyAt(bu, i) = x;
        buAbove = x;
        vecAbove = (anyGet(l, i) * scL) * anyGet(em, i);
      }
    }
    anySt(BU, k4, lane, bu);
  }
  // BL[k] = BL[k-1] + B[k-1]*vec[k-1];  out[k] = (BL + D[k]*vec[k]) + BU[k]
  float BL = 0.f, vecBelow = 0.f, sum = 0.f;
  for (int k4 = 0; k4 < K4; ++k4) {
    const float4 l = anyLd(last, k4, lane), em = e[k4], bu = anyLd(BU, k4, lane);
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = 0; i < 4; ++i) {
      const int k = 4 * k4 + i;
      if (k < K) {
        const float vk = (anyGet(l, i) * scL) * anyGet(em, i);
        if (k) {
          BL = BL + B[k - 1] * vecBelow;
        }
        const float x = BL + D[k] * vk + anyGet(bu, i);
        anyAt(o, i) = x;
        sum = sum + x;
        vecBelow = vk;
      }
    }
    anySt(out, k4, lane, o);
  }
  anySetScale(out, K4, lane, normalise ? 1.0f / sum : 1.0f);
}

// HMM::getNextAlphaBatched (HMM.cpp:787-830) + the scaling sum: out = alpha of the site after prev's (values un-scaled,
// scale as above).  e: this lane's emission row of the NEW site; aC: temporary (suffix sums of prev).
__device__ __forceinline__ void anyAlphaStep(const int K, const float4* e, const float* D, const float* B, const float* U,
                                             const float* cR, const float4* prev, float4* out, float4* aC,
                                             const bool normalise, const int lane)
{
  const int K4 = (K + 3) >> 2;
  const float scP = anyScaleOf(prev, K4, lane);
  // alphaC[K-1] = prev[K-1]; alphaC[k] = alphaC[k+1] + prev[k]
  float above = 0.f;
  for (int k4 = K4 - 1; k4 >= 0; --k4) {
    const float4 pv = anyLd(prev, k4, lane);
    float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = 3; i >= 0; --i) {
      const int k = 4 * k4 + i;
      if (k < K) {
        const float pk = anyGet(pv, i) * scP;
        const float x = (k == K - 1) ? pk : above + pk;
        anyAt(c, i) = x;
        above = x;
      }
    }
    anySt(aC, k4, lane, c);
  }
  float AU = 0.f, prevBelow = 0.f, sum = 0.f;
  for (int k4 = 0; k4 < K4; ++k4) {
    const float4 pv = anyLd(prev, k4, lane), c = anyLd(aC, k4, lane), em = e[k4];
    // alphaC[k+1] of this block's last state sits in the next block
    const float4 cNext = (k4 + 1 < K4) ? anyLd(aC, k4 + 1, lane) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = 0; i < 4; ++i) {
      const int k = 4 * k4 + i;
      if (k < K) {
        const float pk = anyGet(pv, i) * scP;
        if (k) {
          AU = U[k - 1] * prevBelow + cR[k - 1] * AU;
        }
        float term = AU + D[k] * pk;
        if (k < K - 1) {
          const float cAbove = (i < 3) ? anyGet(c, i + 1) : cNext.x;
          term = term + B[k] * cAbove;
        }
        const float x = anyGet(em, i) * term;
        anyAt(o, i) = x;
        sum = sum + x;
        prevBelow = pk;
      }
    }
    anySt(out, k4, lane, o);
  }
  anySetScale(out, K4, lane, normalise ? 1.0f / sum : 1.0f);
}

template <int MODE, bool TRACK, bool SEQ>
__global__ __launch_bounds__(kWave) void decode_kernel_any(const KParams p)
{
  static_assert(!TRACK || MODE == kModeIbd, "segment ages belong to the IBD scan");
  const int lane = threadIdx.x;
  const int K = p.K, KP = p.KP;
  const int K4 = (K + 3) >> 2, E4 = KP >> 2;
  constexpr int NC = SEQ ? 4 : 3;
  const size_t vecF4 = (size_t)(K4 + 1) * kWave; // float4 per row: the values and the scale (the host plans K4 + 1)
  const int C = p.chunk;
  float4* const chunkbuf = p.ws + (size_t)blockIdx.x * p.wsSlot;
  float4* const ckpt = chunkbuf + (size_t)p.chunkRows * vecF4; // [maxChunks + 2]
  float4* const extra = ckpt + (size_t)(p.maxChunks + 2) * vecF4;
  float4* const rowA0 = extra;             // alpha, two rows in turn
  float4* const rowA1 = extra + vecF4;
  float4* const rowB0 = extra + 2 * vecF4; // beta of pass B, two rows in turn
  float4* const rowB1 = extra + 3 * vecF4;
  float4* const tmp = extra + 4 * vecF4;   // BU / alphaC of the step under way
  float4* const post = extra + 5 * vecF4;  // alpha*beta of the current site; its scale slot holds 1 / (their sum)
  float4* const sps = extra + 6 * vecF4;   // per-state posterior sums of the open segment (TRACK)
  float4* const half = extra + 7 * vecF4;  // SEQ: beta after the half-step across the gap
  __shared__ unsigned char clsLds[kWave];

  for (unsigned round = 0;; ++round) {
    unsigned g = 0;
    if (MODE == kModeSums) { // one batch per wave and launch, its groups in turn (decode_kernel, same place)
      if (p.batchFirst) {
        g = p.batchFirst[p.groupBase + blockIdx.x] + round;
        if (g >= p.batchFirst[p.groupBase + blockIdx.x + 1]) {
          g = (unsigned)p.nGroups;
        }
      } else {
        g = round == 0 ? (unsigned)p.groupBase + blockIdx.x : (unsigned)p.nGroups;
      }
    } else {
      if (lane == 0) {
        g = atomicAdd(&p.counters[p.groupBase], 1u);
      }
      g = __builtin_amdgcn_readfirstlane(g);
    }
    if (g >= (unsigned)p.nGroups) {
      break;
    }
    const fsmc_group grp = p.groups[g];
    const int nPairsInGroup = (int)grp.n_pairs;
    const int from = (int)grp.from, to = (int)grp.to, scanFrom = (int)grp.scan_from;
    const int aEnd = (MODE == kModeIbd) ? (int)grp.scan_to : to; // the alpha sweep stops here
    const bool valid = lane < nPairsInGroup;
    const unsigned pairIdx = grp.first_pair + (valid ? (unsigned)lane : 0u);
    const fsmc_pair pr = p.pairs[pairIdx];
    const unsigned long long* hapA = p.haps + (size_t)pr.hap_a * p.W;
    const unsigned long long* hapB = p.haps + (size_t)pr.hap_b * p.W;
    auto obsClass = [&](const int q) -> int { // 0 het, 1 hom major, 2 hom minor (HMM.cpp:647-652)
      const unsigned long long wa = hapA[q >> 6], wb = hapB[q >> 6];
      const int x = (int)(((wa ^ wb) >> (q & 63)) & 1ull), t = (int)(((wa & wb) >> (q & 63)) & 1ull);
      return x ? 0 : 1 + t;
    };
    auto emisRow = [&](const int q, const int cls) -> const float4* {
      return p.emis3 + ((size_t)q * NC + cls) * E4;
    };
    auto tableRow = [&](const float* t, const int row) -> const float* { return t + (size_t)row * KP; };
    // the half-step of sequence mode across the gap in front of site q, backward: un-normalised (HMM.cpp:915-919)
    auto betaGap = [&](const int q, const float4* in, float4* out) {
      const int rg = p.rowGapB[q];
      anyBetaStep(K, emisRow(q, 3), tableRow(p.D, rg), tableRow(p.B, rg), tableRow(p.U, rg), tableRow(p.RR, rg), in, out,
                  tmp, false, lane);
    };
    // the step out of site q, backward, normalised (array mode: THE step; sequence mode: behind the half-step)
    auto betaSite = [&](const int q, const float4* in, float4* out) {
      const int r = SEQ ? p.rowSiteB[q] : p.stepRow[q];
      anyBetaStep(K, emisRow(q, obsClass(q)), tableRow(p.D, r), tableRow(p.B, r), tableRow(p.U, r), tableRow(p.RR, r), in,
                  out, tmp, true, lane);
    };
    // beta of site pos from beta of site pos + 1: last -> out (out != last)
    auto betaInto = [&](const int pos, const float4* last, float4* out) {
      if constexpr (SEQ) {
        betaGap(pos + 1, last, half);
        betaSite(pos + 1, half, out);
      } else {
        betaSite(pos + 1, last, out);
      }
    };
    auto betaInit = [&](float4* out) { // ones; scale 1 / (K sequential adds) (HMM.cpp:887-897)
      float sum = 0.f;
      for (int k4 = 0; k4 < K4; ++k4) {
        float4 v;
        for (int i = 0; i < 4; ++i) {
          const bool real = 4 * k4 + i < K;
          anyAt(v, i) = real ? 1.0f : 0.f;
          if (real) {
            sum = sum + 1.0f;
          }
        }
        anySt(out, k4, lane, v);
      }
      anySetScale(out, K4, lane, 1.0f / sum);
    };

    const int nA = aEnd - from;
    const int nChunks = nA > 0 ? (nA + C - 1) / C : 0;
    // ---- pass B: beta from the window's last site down to the lowest site a chunk starts its rebuild from.
    // Checkpoints: ckpt[j] = beta of the first site of chunk j (j = 1 .. nChunks-1); ckpt[nChunks] = beta of site aEnd when
    // the alpha sweep ends before the window does.
    {
      const int stop = nChunks > 1 ? from + C : aEnd; // (nothing to keep when one chunk reaches the window's end)
      if (nChunks > 0 && stop < to) {
        float4* bCur = rowB0;
        float4* bNxt = rowB1;
        betaInit(bCur);
        for (int pos = to - 1;; --pos) {
          int slot = -1;
          if (pos == aEnd) {
            slot = nChunks;
          } else if (pos < aEnd && pos > from && (pos - from) % C == 0) {
            slot = (pos - from) / C;
          }
          if (slot >= 0) {
            anyCopy(ckpt + (size_t)slot * vecF4, bCur, K4, lane);
          }
          if (pos <= stop) {
            break;
          }
          betaInto(pos - 1, bCur, bNxt);
          float4* t = bCur;
          bCur = bNxt;
          bNxt = t;
        }
      }
    }

    // ---- pass A, chunk by chunk
    float4* aCur = rowA0;
    float4* aNxt = rowA1;
    int cur = 4, segStart = 0; // IBD scan: the open segment's level (4 = none) and first site
    float acc = 0.f;
    for (int j = 0; j < nChunks; ++j) {
      const int lo = from + j * C;
      const int hi = (lo + C < aEnd) ? lo + C : aEnd;
      // rebuild: the beta rows of sites lo .. hi-1 into the chunk buffer, from the window's end or from a checkpoint
      {
        const float4* start = nullptr; // beta of site hi, when the chain starts from a checkpoint
        if (hi != to) {
          start = ckpt + (size_t)(hi == aEnd ? nChunks : j + 1) * vecF4;
        }
        if constexpr (SEQ) {
          // What the reference's beta buffer holds for a site in sequence mode is beta AFTER the half-step across the gap
          // (lastComputedBeta = previousBeta copies it over the site's row, HMM.cpp:915-925) -- un-normalised; only the
          // window's first site keeps its own beta.  The chunk buffer therefore keeps the half-step rows, and the chain
          // runs  beta[s] -> half-step (kept) -> step out of site s = beta[s-1].
          float4* bs = rowB0; // beta of the current site
          int s;
          if (!start) {
            betaInit(bs);
            s = to - 1;
          } else {
            betaInto(hi - 1, start, bs); // (through site hi's half-step, which belongs to the next chunk)
            s = hi - 1;
          }
          for (; s >= lo; --s) {
            float4* out = chunkbuf + (size_t)(s - lo) * vecF4;
            if (s == from) {
              anyCopy(out, bs, K4, lane);
              break;
            }
            betaGap(s, bs, out);
            if (s > lo) {
              betaSite(s, out, bs);
            }
          }
        } else {
          int pos;
          const float4* last;
          if (!start) {
            float4* top = chunkbuf + (size_t)(to - 1 - lo) * vecF4;
            betaInit(top);
            last = top;
            pos = to - 2;
          } else {
            last = start;
            pos = hi - 1;
          }
          for (; pos >= lo; --pos) {
            float4* out = chunkbuf + (size_t)(pos - lo) * vecF4;
            betaInto(pos, last, out);
            last = out;
          }
        }
      }
      // alpha sweep through the chunk
      for (int pos = lo; pos < hi; ++pos) {
        const int c = obsClass(pos);
        if (pos == from) {
          // alpha of the window's first site: pi * emission; scale 1 / sum (HMM.cpp:736-747)
          const float4* e = emisRow(pos, c);
          float sum = 0.f;
          for (int k4 = 0; k4 < K4; ++k4) {
            const float4 em = e[k4];
            float4 v;
            for (int i = 0; i < 4; ++i) {
              const int k = 4 * k4 + i;
              anyAt(v, i) = (k < K) ? p.pi[k] * anyGet(em, i) : 0.f;
              if (k < K) {
                sum = sum + anyAt(v, i);
              }
            }
            anySt(aCur, k4, lane, v);
          }
          anySetScale(aCur, K4, lane, 1.0f / sum);
        } else {
          const int r = p.stepRow[pos]; // (sequence mode: the site step; aNxt holds alpha of site pos - 1 after the
                                        //  half-step across the gap, computed at that site, below)
          if constexpr (SEQ) {
            anyAlphaStep(K, emisRow(pos, c), tableRow(p.D, r), tableRow(p.B, r), tableRow(p.U, r), p.cR, aNxt, aCur, tmp,
                         true, lane);
          } else {
            anyAlphaStep(K, emisRow(pos, c), tableRow(p.D, r), tableRow(p.B, r), tableRow(p.U, r), p.cR, aCur, aNxt, tmp,
                         true, lane);
            float4* t = aCur;
            aCur = aNxt;
            aNxt = t;
          }
        }
        // What the reference's alpha buffer holds for a site in sequence mode: alpha after the un-normalised half-step
        // across the gap to the next site (previousAlpha = nextAlpha copies it over the site's row, HMM.cpp:760-770); the
        // window's last site keeps its alpha.  The half-step is also what the next site's step starts from.
        const float4* aUse = aCur;
        if constexpr (SEQ) {
          if (pos < to - 1) {
            const int rg = p.rowGapF[pos + 1];
            anyAlphaStep(K, emisRow(pos + 1, 3), tableRow(p.D, rg), tableRow(p.B, rg), tableRow(p.U, rg), p.cR, aCur, aNxt,
                         tmp, false, lane);
            aUse = aNxt;
          }
        }
        // combine with beta of this site and normalise (HMM.cpp:669-692): post = alpha*beta, its scale 1 / sum
        const float4* bRow = chunkbuf + (size_t)(pos - lo) * vecF4;
        const float scA = anyScaleOf(aUse, K4, lane), scB = anyScaleOf(bRow, K4, lane);
        float sumq = 0.f;
        for (int k4 = 0; k4 < K4; ++k4) {
          const float4 a = anyLd(aUse, k4, lane), b = anyLd(bRow, k4, lane);
          float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
          for (int i = 0; i < 4 && 4 * k4 + i < K; ++i) {
            const float x = (anyGet(a, i) * scA) * (anyGet(b, i) * scB);
            anyAt(q, i) = x;
            sumq = sumq + x;
          }
          anySt(post, k4, lane, q);
        }
        const float cq = 1.0f / sumq;

        if (MODE == kModeDump) {
          float* out = p.dumpOut + p.dumpOffsets[g] + (size_t)(pos - from) * K * kWave + lane;
          for (int k4 = 0; k4 < K4; ++k4) {
            const float4 q = anyLd(post, k4, lane);
            for (int i = 0; i < 4 && 4 * k4 + i < K; ++i) {
              out[(size_t)(4 * k4 + i) * kWave] = valid ? anyGet(q, i) * cq : 0.f;
            }
          }
        }

        if (MODE == kModePerPair) {
          // HMM::writePerPairOutput (HMM.cpp:1378-1409): mean = sum_k post*E[t_k], MAP = first strictly larger posterior
          float mean = 0.f, best = 0.f;
          int arg = 0;
          for (int k4 = 0; k4 < K4; ++k4) {
            const float4 q = anyLd(post, k4, lane);
            for (int i = 0; i < 4 && 4 * k4 + i < K; ++i) {
              const float pk = anyGet(q, i) * cq;
              mean = mean + pk * p.expCoal[4 * k4 + i];
              if (best < pk) {
                arg = 4 * k4 + i;
                best = pk;
              }
            }
          }
          if (valid) {
            if (p.ppMean) p.ppMean[(size_t)pairIdx * p.S + pos] = mean;
            if (p.ppMap) p.ppMap[(size_t)pairIdx * p.S + pos] = arg;
          }
        }

        if (MODE == kModeSums) {
          // HMM::augmentSumOverPairs (HMM.cpp:1052-1081): per site and state the batch's posteriors summed over its pairs
          // in batch order; lane j takes the states j, j + 64, ... and reads the pairs' values from the posterior row
          anySetScale(post, K4, lane, cq);
          clsLds[lane] = (unsigned char)c;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          __builtin_amdgcn_s_barrier();
          const float* pf = reinterpret_cast<const float*>(post);
          for (int kk = lane; kk < K; kk += kWave) {
            float* a = p.sums + (size_t)blockIdx.x * p.sumsSlot + (size_t)pos * K + kk;
            float s = 0.f, s00 = 0.f, s01 = 0.f, s11 = 0.f;
            if (round > 0) {
              if (p.flags & FSMC_WANT_SUMS) s = a[0];
              if (p.flags & FSMC_WANT_MAJOR_MINOR_SUMS) {
                s00 = a[p.sumsPlane];
                s01 = a[2 * p.sumsPlane];
                s11 = a[3 * p.sumsPlane];
              }
            }
            for (int v = 0; v < nPairsInGroup; ++v) {
              const float q = pf[((size_t)(kk >> 2) * kWave + v) * 4 + (kk & 3)] * pf[((size_t)K4 * kWave + v) * 4];
              s = s + q;
              if (p.flags & FSMC_WANT_MAJOR_MINOR_SUMS) {
                const int cv = clsLds[v];
                if (cv == 2) {
                  s11 = s11 + q;
                } else if (cv == 1) {
                  s00 = s00 + q;
                } else {
                  s01 = s01 + q;
                }
              }
            }
            if (p.flags & FSMC_WANT_SUMS) a[0] = s;
            if (p.flags & FSMC_WANT_MAJOR_MINOR_SUMS) {
              a[p.sumsPlane] = s00;
              a[2 * p.sumsPlane] = s01;
              a[3 * p.sumsPlane] = s11;
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          __builtin_amdgcn_s_barrier();
        }

        if (MODE == kModeIbd && pos >= scanFrom) {
          // sum over the states below the threshold, k ascending from 0.f (HMM.cpp:1207-1224; with segment ages the
          // reference's loop runs to the age threshold and adds the states below the state threshold on the way)
          const unsigned nSum = TRACK ? (p.stateThr < p.ageThr ? p.stateThr : p.ageThr) : p.stateThr;
          float s = 0.f;
          for (int k4 = 0; 4 * k4 < (int)nSum; ++k4) {
            const float4 q = anyLd(post, k4, lane);
            for (int i = 0; i < 4 && 4 * k4 + i < (int)nSum; ++i) {
              s = s + anyGet(q, i) * cq;
            }
          }
          const int level = s >= p.thr[0] ? 0 : s >= p.thr[1] ? 1 : s >= p.thr[2] ? 2 : s >= p.thr[3] ? 3 : 4;
          auto emit = [&](const int s0, const int s1) {
            const unsigned idx = atomicAdd(&p.counters[1], 1u);
            float mean = 0.f, mapv = 0.f;
            if constexpr (TRACK) {
              segment_ages(K, p.ageThr, sps + lane, (cfloat_p)p.pi, (cfloat_p)p.expT, (p.flags & FSMC_WANT_MEAN) != 0,
                           (p.flags & FSMC_WANT_MAP) != 0, mean, mapv);
            }
            if (idx < p.recCap) {
              fsmc_ibd_record r;
              r.pair = pairIdx;
              r.start = s0;
              r.end = s1;
              r.prob = acc;
              r.post_mean = mean;
              r.map = mapv;
              p.recs[idx] = r;
            }
          };
          if (valid && cur != 4 && level != cur) { // a change of level closes the open segment at pos - 1
            emit(segStart, pos - 1);
          }
          const bool opening = level != 4 && level != cur;
          if constexpr (TRACK) {
            if (level != 4) { // sum_posterior_per_state (HMM.cpp:1212-1229) of the states below the age threshold
              for (int k4 = 0; 4 * k4 < (int)p.ageThr && k4 < K4; ++k4) {
                const float4 q = anyLd(post, k4, lane);
                float4 sv = opening ? make_float4(0.f, 0.f, 0.f, 0.f) : anyLd(sps, k4, lane);
                sv.x = sv.x + q.x * cq;
                sv.y = sv.y + q.y * cq;
                sv.z = sv.z + q.z * cq;
                sv.w = sv.w + q.w * cq;
                anySt(sps, k4, lane, sv);
              }
            }
          }
          acc = (level == 4) ? 0.f : (opening ? s : acc + s);
          if (opening) {
            segStart = pos;
          }
          cur = level;
          if (pos == aEnd - 1 && valid && cur != 4) {
            emit(segStart, pos);
          }
        }
      }
    }
  }
}

} // namespace fsmc
