// fsmc_kernels.h -- device side of libfastsmc_hip.so: the pairwise-HMM decode kernel for gfx950.
//
// Mapping: lane = haplotype pair, one wavefront = one reference batch of <= 64 pairs
// that share a decode window; the K states are walked sequentially by every lane, so every sum and
// recurrence is evaluated in the reference's order (NO_SSE variant) and results are
// bit-identical to the CPU path.  No FMA contraction (-ffp-contract=off), IEEE division.
//
// Per wave, per group:
//   pass B : beta sweep, site to-1 down to from; keeps only checkpoints every `chunk` sites
//            (or every beta when the whole window fits the workspace: single-chunk mode)
//   pass A : for each chunk, ascending: recompute the chunk's betas from the checkpoint into the
//            wave's private chunk buffer (HBM), then the alpha sweep through the chunk, fusing
//            combine/normalise and the posterior consumer (IBD scan / dump / per-pair / sums).
// Algorithmic HBM traffic: one 4*K-byte beta row written and read once per pair-site (8K + 0.25 B).
//
// Reference statements this follows (ASMC_SRC/SRC): HMM.cpp:725-784 + 787-830 (forward),
// 882-940 + 943-1016 (backward), 669-692 (combine), HmmUtils.cpp:102-151 (scaling),
// HMM.cpp:1179-1357 (IBD scan), 1087-1107 (segment age estimates).
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/fastsmc_hip.h"

namespace fsmc
{

constexpr int kWave = 64;

// Read-only, wave-uniform model data is addressed through the constant address space: the compiler
// may then use scalar (SMEM) loads into SGPRs instead of per-lane vector loads + v_readfirstlane.
// Valid because nothing in a launch ever writes these buffers.
typedef const float __attribute__((address_space(4))) * cfloat_p;
typedef const int __attribute__((address_space(4))) * cint_p;
typedef const unsigned __attribute__((address_space(4))) * cuint_p;
constexpr int kMaxGenericK = 256; // upper bound on K for the generic (runtime-K) kernel

enum Mode : int { kModeIbd = 0, kModeDump = 1, kModePerPair = 2, kModeSums = 3 };

struct KParams {
  int K;       // states
  int KP;      // padded row stride of the tables (multiple of 4)
  int S;       // sites
  int W;       // 64-bit words per haplotype row
  int nGroups;
  int chunk;   // sites per chunk (C)
  int maxChunks;
  unsigned flags;
  const float* pi;    // [KP]
  const float* cR;    // [KP]
  const float* expT;  // [KP]
  const float* D;     // [rows][KP]
  const float* B;
  const float* U;
  const float* RR;
  const int* stepRow; // [S]
  const float4* emis3; // [S][3][KP/4]: emission rows for obs class het / hom-major / hom-minor
  const unsigned long long* haps; // [nHaps][W]
  const fsmc_pair* pairs;
  const fsmc_group* groups;
  unsigned* counters; // [0] group queue head, [1] IBD record count
  float4* ws;         // workspace, wsSlot float4 per resident wave
  size_t wsSlot;
  unsigned stateThr, ageThr;
  float thr[4];       // {1000,100,10,1} * probabilityThreshold, evaluated in fp32 (HMM.cpp:1226...)
  fsmc_ibd_record* recs;
  unsigned recCap;
  float* dumpOut;             // kModeDump
  const size_t* dumpOffsets;  // [nGroups] float offsets
  float* ppMean;              // kModePerPair: [nPairs][S]
  int* ppMap;                 // kModePerPair: [nPairs][S]
  const float* expCoal;       // kModePerPair: [KP]
  float* sums;                // kModeSums: per-slot accumulators [slots][S][K] (+ 00/01/11 planes)
  size_t sumsPlane;           // floats per plane per slot
};

// ---------------------------------------------------------------------------------------------
// One step of the backward recursion for one pair (HMM.cpp:957-1016, NO_SSE association).
// b: beta of site pos+1 (scaled) on entry, beta of site pos (scaled) on exit.  w: scratch.
// e: this lane's emission row for site pos+1 (LDS).  Dr/Br/Ur/RRr: wave-uniform table rows.
template <int KT, int KA>
__device__ __forceinline__ void beta_step(const int K, float (&b)[KA], float (&w)[KA], cfloat_p Dr, cfloat_p Br,
                                          cfloat_p Ur, cfloat_p RRr, const float4* e)
{
  const int K4 = (K + 3) >> 2;
#pragma unroll
  for (int k4 = 0; k4 < K4; ++k4) {
    const float4 ev = e[k4];
    if (4 * k4 + 0 < K) b[4 * k4 + 0] = b[4 * k4 + 0] * ev.x;
    if (4 * k4 + 1 < K) b[4 * k4 + 1] = b[4 * k4 + 1] * ev.y;
    if (4 * k4 + 2 < K) b[4 * k4 + 2] = b[4 * k4 + 2] * ev.z;
    if (4 * k4 + 3 < K) b[4 * k4 + 3] = b[4 * k4 + 3] * ev.w;
  }
  w[K - 1] = 0.f;
#pragma unroll
  for (int k = K - 2; k >= 0; --k) {
    w[k] = Ur[k] * b[k + 1] + RRr[k] * w[k + 1];
  }
  float BL = 0.f;
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if (k) {
      BL = BL + Br[k - 1] * b[k - 1];
    }
    w[k] = (BL + Dr[k] * b[k]) + w[k];
    sum = sum + w[k];
  }
  const float c = 1.0f / sum;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    b[k] = w[k] * c;
  }
}

// One step of the forward recursion (HMM.cpp:799-830) followed by the per-site scaling
// (HmmUtils.cpp:102-151).  a: alpha of site pos-1 on entry, of site pos on exit.
template <int KT, int KA>
__device__ __forceinline__ void alpha_step(const int K, float (&a)[KA], float (&w)[KA], cfloat_p Dr, cfloat_p Br,
                                           cfloat_p Ur, cfloat_p cR, const float4* e)
{
  w[K - 1] = a[K - 1];
#pragma unroll
  for (int k = K - 2; k >= 0; --k) {
    w[k] = w[k + 1] + a[k];
  }
  float AU = 0.f;
  float sum = 0.f;
  float4 ev = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if ((k & 3) == 0) {
      ev = e[k >> 2];
    }
    if (k) {
      AU = Ur[k - 1] * a[k - 1] + cR[k - 1] * AU;
    }
    float term = AU + Dr[k] * a[k];
    if (k < K - 1) {
      term = term + Br[k] * w[k + 1];
    }
    const float em = (k & 3) == 0 ? ev.x : (k & 3) == 1 ? ev.y : (k & 3) == 2 ? ev.z : ev.w;
    w[k] = em * term;
    sum = sum + w[k];
  }
  const float c = 1.0f / sum;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    a[k] = w[k] * c;
  }
}

// alpha at the first site of the window: pi * emission, scaled (HMM.cpp:736-747).
template <int KT, int KA>
__device__ __forceinline__ void alpha_init(const int K, float (&a)[KA], cfloat_p pi, const float4* e)
{
  float sum = 0.f;
  float4 ev = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if ((k & 3) == 0) {
      ev = e[k >> 2];
    }
    const float em = (k & 3) == 0 ? ev.x : (k & 3) == 1 ? ev.y : (k & 3) == 2 ? ev.z : ev.w;
    a[k] = pi[k] * em;
    sum = sum + a[k];
  }
  const float c = 1.0f / sum;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    a[k] = a[k] * c;
  }
}

// beta at the last site of the window: all ones, scaled (HMM.cpp:887-897).
template <int KT, int KA> __device__ __forceinline__ void beta_init(const int K, float (&b)[KA])
{
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    b[k] = 1.0f;
    sum = sum + b[k];
  }
  const float c = 1.0f / sum;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    b[k] = b[k] * c;
  }
}

// A K-vector of one wave lives in HBM as [K/4][64 lanes] float4: one coalesced 1-KiB row per
// group of four states (global_store/load_dwordx4).
template <int KT, int KA> __device__ __forceinline__ void store_vec(const int K, float4* dst, const float (&v)[KA])
{
  const int K4 = (K + 3) >> 2;
#pragma unroll
  for (int k4 = 0; k4 < K4; ++k4) {
    float4 o;
    o.x = v[4 * k4];
    o.y = (4 * k4 + 1 < K) ? v[4 * k4 + 1] : 0.f;
    o.z = (4 * k4 + 2 < K) ? v[4 * k4 + 2] : 0.f;
    o.w = (4 * k4 + 3 < K) ? v[4 * k4 + 3] : 0.f;
    dst[(size_t)k4 * kWave] = o;
  }
}

template <int KT, int KA> __device__ __forceinline__ void load_vec(const int K, const float4* src, float (&v)[KA])
{
  const int K4 = (K + 3) >> 2;
#pragma unroll
  for (int k4 = 0; k4 < K4; ++k4) {
    const float4 o = src[(size_t)k4 * kWave];
    v[4 * k4] = o.x;
    if (4 * k4 + 1 < K) v[4 * k4 + 1] = o.y;
    if (4 * k4 + 2 < K) v[4 * k4 + 2] = o.z;
    if (4 * k4 + 3 < K) v[4 * k4 + 3] = o.w;
  }
}

// Segment age estimates from the per-state posterior sums of a segment
// (HMM::getPosteriorMean, HMM.cpp:1087-1097; HMM::getMAP, 1099-1107).
template <int KT, int KA>
__device__ __forceinline__ void segment_ages(const int K, const unsigned nAge, const float (&sps)[KA], cfloat_p pi,
                                             cfloat_p expT, const bool wantMean, const bool wantMap, float& mean,
                                             float& mapv)
{
  mean = 0.f;
  mapv = 0.f;
  if (wantMean) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if ((unsigned)k < nAge) acc = acc + sps[k];
    }
    const float norm = 1.f / acc;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if ((unsigned)k < nAge) mean = mean + (norm * sps[k]) * expT[k];
    }
  }
  if (wantMap) {
    float best = 0.f;
    float bestT = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if ((unsigned)k < nAge) {
        const float r = sps[k] / pi[k];
        if (k == 0 || best < r) {
          best = r;
          bestT = expT[k];
        }
      }
    }
    mapv = bestT;
  }
}

template <int KT, int MODE, bool TRACK>
__global__ __launch_bounds__(kWave, 2) void decode_kernel(const KParams p)
{
  constexpr int KA = KT > 0 ? KT : kMaxGenericK;
  constexpr int K4A = (KA + 3) / 4;
  const int K = KT > 0 ? KT : p.K;
  const int K4 = (K + 3) >> 2;
  const int KP = p.KP;

  __shared__ float4 emisLds[2][3 * K4A];

  const int lane = threadIdx.x;
  const cfloat_p tD = (cfloat_p)p.D, tB = (cfloat_p)p.B, tU = (cfloat_p)p.U, tRR = (cfloat_p)p.RR;
  const cfloat_p tPi = (cfloat_p)p.pi, tCR = (cfloat_p)p.cR, tExpT = (cfloat_p)p.expT;
  const cint_p tStepRow = (cint_p)p.stepRow;
  const size_t vecF4 = (size_t)K4 * kWave; // float4 per stored K-vector of a wave
  float4* const chunkbuf = p.ws + (size_t)blockIdx.x * p.wsSlot;
  float4* const ckpt = chunkbuf + (size_t)p.chunk * vecF4;
  float4* const saveA = ckpt + (size_t)(p.maxChunks + 2) * vecF4;
  float4* const saveS = saveA + vecF4;
  const int C = p.chunk;

  for (;;) {
    unsigned g = 0;
    if (lane == 0) {
      g = atomicAdd(&p.counters[0], 1u);
    }
    g = __builtin_amdgcn_readfirstlane(g);
    if (g >= (unsigned)p.nGroups) {
      break;
    }
    const cuint_p gw = (cuint_p)(p.groups + g);
    fsmc_group grp;
    grp.first_pair = gw[0];
    grp.n_pairs = gw[1];
    grp.from = gw[2];
    grp.to = gw[3];
    grp.scan_from = gw[4];
    grp.scan_to = gw[5];
    const int from = (int)grp.from;
    const int to = (int)grp.to;
    const int scanFrom = (int)grp.scan_from;
    const int aEnd = (MODE == kModeIbd) ? (int)grp.scan_to : to; // the alpha sweep stops here
    const bool valid = lane < (int)grp.n_pairs;
    const unsigned pairIdx = grp.first_pair + (valid ? (unsigned)lane : 0u);
    const fsmc_pair pr = p.pairs[pairIdx];
    const unsigned long long* rowA = p.haps + (size_t)pr.hap_a * p.W;
    const unsigned long long* rowB = p.haps + (size_t)pr.hap_b * p.W;

    const int nA = aEnd - from;
    const int nChunks = (nA + C - 1) / C;
    const bool single = nChunks <= 1;

    int wordIdx = -1;
    unsigned long long xw = 0, aw = 0;
    // observation class of this lane's pair at site q: 0 het, 1 hom major, 2 hom minor
    // (obsIsZero / obsIsTwo of HMM.cpp:647-652 folded into a row select)
    auto obsClass = [&](const int q) -> int {
      const int wi = q >> 6;
      if (wi != wordIdx) {
        const unsigned long long wa = rowA[wi];
        const unsigned long long wb = rowB[wi];
        xw = wa ^ wb;
        aw = wa & wb;
        wordIdx = wi;
      }
      const int bit = q & 63;
      const int x = (int)((xw >> bit) & 1ull);
      const int t = (int)((aw >> bit) & 1ull);
      return x ? 0 : 1 + t;
    };
    auto prefetchEmis = [&](const int q) -> float4 {
      float4 v = {0.f, 0.f, 0.f, 0.f};
      if (lane < 3 * K4) {
        v = p.emis3[(size_t)q * (3 * K4) + lane];
      }
      return v;
    };
    auto commitEmis = [&](const int q, const float4 v) {
      if (lane < 3 * K4) {
        emisLds[q & 1][lane] = v;
      }
      __builtin_amdgcn_wave_barrier();
    };

    float w[KA];

    // ------------------------------------------------------------------ pass B
    {
      float b[KA];
      beta_init<KT, KA>(K, b);
      auto afterBeta = [&](const int pos) {
        if (single) {
          if (pos < aEnd) {
            store_vec<KT, KA>(K, chunkbuf + (size_t)(pos - from) * vecF4 + lane, b);
          }
        } else {
          const int rel = pos - from;
          if (rel > 0 && pos <= aEnd && (rel % C == 0 || pos == aEnd)) {
            const int j = (pos == aEnd) ? nChunks : rel / C;
            store_vec<KT, KA>(K, ckpt + (size_t)j * vecF4 + lane, b);
          }
        }
      };
      afterBeta(to - 1);
      float4 ev = {0.f, 0.f, 0.f, 0.f};
      if (to - 2 >= from) {
        ev = prefetchEmis(to - 1);
      }
      for (int pos = to - 2; pos >= from; --pos) {
        const int q = pos + 1;
        commitEmis(q, ev);
        if (pos - 1 >= from) {
          ev = prefetchEmis(q - 1);
        }
        const int c = obsClass(q);
        const size_t row = (size_t)tStepRow[q] * KP;
        beta_step<KT, KA>(K, b, w, tD + row, tB + row, tU + row, tRR + row, &emisLds[q & 1][c * K4]);
        afterBeta(pos);
      }
    }

    // ------------------------------------------------------------------ pass A
    int cur = 4;      // open threshold level (0..3) or 4 = none
    int segStart = 0; // first site of the open segment
    float acc = 0.f;  // posteriorIBD
    float a[KA];
    float sps[TRACK ? KA : 1];
    if constexpr (TRACK) {
#pragma unroll
      for (int k = 0; k < K; ++k) sps[k] = 0.f;
    }

    auto emit = [&](const int s0, const int s1) {
      const unsigned idx = atomicAdd(&p.counters[1], 1u);
      float mean = 0.f, mapv = 0.f;
      if constexpr (TRACK) {
        segment_ages<KT, KA>(K, p.ageThr, sps, tPi, tExpT, (p.flags & FSMC_WANT_MEAN) != 0,
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

    for (int j = 0; j < (nChunks > 0 ? nChunks : 0); ++j) {
      const int lo = from + j * C;
      const int hi = (lo + C < aEnd) ? lo + C : aEnd;
      if (!single) {
        // park the carried alpha (and per-state sums) while the chunk's betas are rebuilt
        if (j > 0) {
          store_vec<KT, KA>(K, saveA + lane, a);
          if constexpr (TRACK) store_vec<KT, KA>(K, saveS + lane, sps);
        }
        {
          float b[KA];
          int pos;
          if (hi == to) {
            beta_init<KT, KA>(K, b);
            store_vec<KT, KA>(K, chunkbuf + (size_t)(to - 1 - lo) * vecF4 + lane, b);
            pos = to - 2;
          } else {
            load_vec<KT, KA>(K, ckpt + (size_t)(j + 1) * vecF4 + lane, b);
            pos = hi - 1;
          }
          float4 ev = {0.f, 0.f, 0.f, 0.f};
          if (pos >= lo) {
            ev = prefetchEmis(pos + 1);
          }
          for (; pos >= lo; --pos) {
            const int q = pos + 1;
            commitEmis(q, ev);
            if (pos - 1 >= lo) {
              ev = prefetchEmis(q - 1);
            }
            const int c = obsClass(q);
            const size_t row = (size_t)tStepRow[q] * KP;
            beta_step<KT, KA>(K, b, w, tD + row, tB + row, tU + row, tRR + row, &emisLds[q & 1][c * K4]);
            store_vec<KT, KA>(K, chunkbuf + (size_t)(pos - lo) * vecF4 + lane, b);
          }
        }
        if (j > 0) {
          load_vec<KT, KA>(K, saveA + lane, a);
          if constexpr (TRACK) load_vec<KT, KA>(K, saveS + lane, sps);
        }
      }

      float4 ev = prefetchEmis(lo);
      for (int pos = lo; pos < hi; ++pos) {
        commitEmis(pos, ev);
        if (pos + 1 < hi) {
          ev = prefetchEmis(pos + 1);
        }
        const int c = obsClass(pos);
        const float4* e = &emisLds[pos & 1][c * K4];
        if (pos == from) {
          alpha_init<KT, KA>(K, a, tPi, e);
        } else {
          const size_t row = (size_t)tStepRow[pos] * KP;
          alpha_step<KT, KA>(K, a, w, tD + row, tB + row, tU + row, tCR, e);
        }

        // combine with beta of this site and normalise (HMM.cpp:672-691)
        const float4* bsrc = chunkbuf + (size_t)(pos - lo) * vecF4 + lane;
        float sumq = 0.f;
#pragma unroll
        for (int k4 = 0; k4 < K4; ++k4) {
          const float4 bv = bsrc[(size_t)k4 * kWave];
          w[4 * k4] = a[4 * k4] * bv.x;
          sumq = sumq + w[4 * k4];
          if (4 * k4 + 1 < K) {
            w[4 * k4 + 1] = a[4 * k4 + 1] * bv.y;
            sumq = sumq + w[4 * k4 + 1];
          }
          if (4 * k4 + 2 < K) {
            w[4 * k4 + 2] = a[4 * k4 + 2] * bv.z;
            sumq = sumq + w[4 * k4 + 2];
          }
          if (4 * k4 + 3 < K) {
            w[4 * k4 + 3] = a[4 * k4 + 3] * bv.w;
            sumq = sumq + w[4 * k4 + 3];
          }
        }
        const float cq = 1.0f / sumq;

        if (MODE == kModeDump) {
          float* out = p.dumpOut + p.dumpOffsets[g] + (size_t)(pos - from) * K * kWave + lane;
#pragma unroll
          for (int k = 0; k < K; ++k) {
            out[(size_t)k * kWave] = valid ? w[k] * cq : 0.f;
          }
        }

        if (MODE == kModeIbd) {
          if (pos >= scanFrom) {
            // posterior of the states the scan needs
            const unsigned nPost = TRACK ? (p.ageThr > p.stateThr ? p.ageThr : p.stateThr) : p.stateThr;
#pragma unroll
            for (int k4 = 0; k4 < K4; ++k4) {
              if ((unsigned)(4 * k4) < nPost) {
                w[4 * k4] = w[4 * k4] * cq;
                if (4 * k4 + 1 < K) w[4 * k4 + 1] = w[4 * k4 + 1] * cq;
                if (4 * k4 + 2 < K) w[4 * k4 + 2] = w[4 * k4 + 2] * cq;
                if (4 * k4 + 3 < K) w[4 * k4 + 3] = w[4 * k4 + 3] * cq;
              }
            }
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < K; ++k) {
              if ((unsigned)k < p.stateThr) s = s + w[k];
            }
            const int level = s >= p.thr[0] ? 0 : s >= p.thr[1] ? 1 : s >= p.thr[2] ? 2 : s >= p.thr[3] ? 3 : 4;
            // a change of level (or a drop below every threshold) closes the open segment at pos-1
            if (valid && cur != 4 && level != cur) {
              emit(segStart, pos - 1);
            }
            const bool opening = level != 4 && level != cur;
            if constexpr (TRACK) {
              if (level != 4) {
#pragma unroll
                for (int k = 0; k < K; ++k) {
                  if ((unsigned)k < p.ageThr) sps[k] = (opening ? 0.f : sps[k]) + w[k];
                }
              }
            }
            acc = (level == 4) ? 0.f : (opening ? s : acc + s);
            if (opening) {
              segStart = pos;
            }
            cur = level;
            if (pos == aEnd - 1) {
              if (valid && cur != 4) {
                emit(segStart, pos);
              }
            }
          }
        }
      }
    }
  }
}

} // namespace fsmc
