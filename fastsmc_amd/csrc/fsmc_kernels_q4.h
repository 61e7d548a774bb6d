// fsmc_kernels_q4.h -- the decode kernel for wide models (70 <= K <= 256; K = 256 is BASELINE.json config 4).
//
// Three 256-float K-vectors per pair do not fit a lane's 256 registers, and the runtime-K kernel that keeps them in
// scratch memory is latency-bound (1.7 % of the HBM roofline at K = 256).  Here FOUR adjacent lanes share a pair:
// lane 4p+q holds states [KQ*q, KQ*q+KQ) of pair p in registers (KQ = 32, 48 or 64 states per lane), a wavefront
// decodes 16 pairs.
//   * every operation that is not part of a recurrence runs on all 64 lanes at once (each on its own 64 states);
//   * the first-order recurrences (BU, BL, AU, suffix sum, scaling sums) are inherently sequential in k, hence
//     sequential across the four lanes of a pair: they run in four PHASES, quarter after quarter, under the lane
//     mask of the active quarter, and the running value crosses to the next lane by a DPP quad permute.
// Every value is produced by the same IEEE operation on the same operands, in the same order, as in the reference
// (HMM.cpp:799-830, 957-1016, HmmUtils.cpp:102-151) -- results are bit-identical; the price is that a recurrence
// instruction serves 16 pairs instead of 64.
// Operands differ between the quarters of a wave, so they cannot be SGPRs: the four table rows and the three
// emission rows of a site are landed in LDS by LDS-DMA one site ahead (no VGPR staging) and read per lane.
// Beta stride 1, array mode, consumers: IBD scan (with segment ages), posterior dump, per-pair mean / MAP rows.
//
// A model with K < 4*KQ states is padded with GHOST states K .. 4*KQ-1 whose table, emission and prior entries are
// zero (the host pads every row to KP = 4*KQ floats).  Ghost values stay exactly +0 through every operation -- and a
// sum that adds +0 in state order is the same sum -- with one exception: beta'[k] = (BL[k] + D[k]*vec[k]) + BU[k] is
// BL[k] for a ghost, so the backward step multiplies beta' by a 1/0 mask row before the scaling sum (x*1.0f is
// exact).  The boundary cases of the real last state need no special code either: BU[K-1] = U*0 + RR*0 = 0 and the
// forward term's B[K-1]*alphaC[K] = B*0 = 0 are what the reference writes explicitly (HMM.cpp:986-1005, 823-826).
#pragma once

#include <type_traits>

#include "fsmc_kernels.h"

namespace fsmc
{

// Waves per SIMD the compiler must leave room for.  At 64 states per lane the three K-vectors plus operands need
// ~400 registers: with the 256 of two waves per SIMD 150 of them spill to scratch memory; one wave per SIMD gets
// 512 (the rest in AGPRs, no scratch) and measured 14 % faster -- the LDS ring allows only five waves per CU anyway.
#ifndef FSMC_Q4_MINBLOCKS
#define FSMC_Q4_MINBLOCKS(KQ) ((KQ) == 64 ? 1 : 2)
#endif
// The four phases of a recurrence as a loop (one copy of the body, a taken branch per phase) or unrolled (four copies,
// no branch: a taken branch costs an in-order wave ~100 cycles, and a step has ~24 of them).
#if defined(FSMC_Q4_ROLLED_PHASES)
#define FSMC_Q4_PHASE_LOOP _Pragma("nounroll")
#else
#define FSMC_Q4_PHASE_LOOP _Pragma("unroll")
#endif
constexpr int kQ4MaxStates = 64;          // largest KQ: 4 * 64 = 256 states
constexpr int kQuadUp = 0xF9;             // quad_perm [1,2,3,3]: lane q reads lane q+1
constexpr int kQuadDn = 0x90;             // quad_perm [0,0,1,2]: lane q reads lane q-1
constexpr int kQuadB3 = 0xFF;             // every lane of the quad reads lane 3
constexpr int kQuadB0 = 0x00;             // every lane of the quad reads lane 0

template <int CTRL> __device__ __forceinline__ float quadMove(const float v)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
#else
  return v;
#endif
}

__device__ __forceinline__ float f4at(const float4& q, const int i)
{
  return i == 0 ? q.x : i == 1 ? q.y : i == 2 ? q.z : q.w;
}

// Sum of a lane's 64 values continued across the four lanes of a pair in state order: returns the running sum after
// this lane's states in every lane; `total` (all lanes) is the sum over the 256 states (HmmUtils.cpp:121-128 order).
template <int KQ> __device__ __forceinline__ float quadOrderedSum(const float (&v)[KQ], const int qd)
{
  float sOut = 0.f;
FSMC_Q4_PHASE_LOOP
  for (int ph = 0; ph < 4; ++ph) {
    const float c = quadMove<kQuadDn>(sOut);
    if (qd == ph) {
      float s = (ph == 0) ? 0.f : c;
#pragma unroll
      for (int j = 0; j < KQ; ++j) {
        s = s + v[j];
      }
      sOut = s;
    }
  }
  return quadMove<kQuadB3>(sOut);
}

// One backward step.  b: beta of site pos+1 on entry, of site pos on exit.  w, x: scratch.  e: this lane's 16
// emission float4 of site pos+1; rD/rB/rUsh/rRR: this lane's slices of the step's table rows (LDS).
template <int KQ>
__device__ __forceinline__ void beta_step_q4(float (&b)[KQ], float (&w)[KQ], float (&x)[KQ], const float4* e,
                                             const float4* rD, const float4* rB, const float4* rUsh,
                                             const float4* rRR, const float4* mask, const int qd)
{
  // vec[k] = beta[k]*e[k] (kept in b), T[k] = U[k-1]*vec[k] (in x)
#pragma unroll
  for (int j4 = 0; j4 < (KQ / 4); ++j4) {
    const float4 em = e[j4];
    const float4 us = rUsh[j4];
    f32x2 v0 = {b[4 * j4], b[4 * j4 + 1]}, v1 = {b[4 * j4 + 2], b[4 * j4 + 3]};
    const f32x2 e0 = {em.x, em.y}, e1 = {em.z, em.w}, u0 = {us.x, us.y}, u1 = {us.z, us.w};
    v0 = v0 * e0;
    v1 = v1 * e1;
    const f32x2 t0 = u0 * v0, t1 = u1 * v1;
    b[4 * j4] = v0.x;
    b[4 * j4 + 1] = v0.y;
    b[4 * j4 + 2] = v1.x;
    b[4 * j4 + 3] = v1.y;
    x[4 * j4] = t0.x;
    x[4 * j4 + 1] = t0.y;
    x[4 * j4 + 2] = t1.x;
    x[4 * j4 + 3] = t1.y;
  }
  // BU[k] = U[k]*vec[k+1] + RR[k]*BU[k+1], BU[K-1] = 0 (HMM.cpp:986-1005): quarter 3 first
  {
    // this lane's RR values are the same in whichever phase it is active: read them once, ahead of the chain (inside
    // the chain every LDS read would sit on the critical path of a single active quarter)
    float4 rrv[KQ / 4];
#pragma unroll
    for (int j4 = 0; j4 < (KQ / 4); ++j4) {
      rrv[j4] = rRR[j4];
    }
    float tLow = 0.f, buLow = 0.f; // T and BU of this lane's lowest state, for the quarter below
FSMC_Q4_PHASE_LOOP
    for (int ph = 3; ph >= 0; --ph) {
      const float cT = quadMove<kQuadUp>(tLow);
      const float cB = quadMove<kQuadUp>(buLow);
      if (qd == ph) {
        w[KQ - 1] = (ph == 3) ? 0.f : cT + rrv[(KQ / 4) - 1].w * cB;
#pragma unroll
        for (int j4 = (KQ / 4) - 1; j4 >= 0; --j4) {
#pragma unroll
          for (int i = 3; i >= 0; --i) {
            const int j = 4 * j4 + i;
            if (j < KQ - 1) {
              w[j] = x[j + 1] + f4at(rrv[j4], i) * w[j + 1];
            }
          }
        }
        tLow = x[0];
        buLow = w[0];
      }
    }
  }
  // bv = B*vec (in x), then BL as running sums: x[j] = BL of state j+1 (HMM.cpp:1008-1016)
#pragma unroll
  for (int j4 = 0; j4 < (KQ / 4); ++j4) {
    const float4 bt = rB[j4];
    const f32x2 v0 = {b[4 * j4], b[4 * j4 + 1]}, v1 = {b[4 * j4 + 2], b[4 * j4 + 3]};
    const f32x2 c0 = {bt.x, bt.y}, c1 = {bt.z, bt.w};
    const f32x2 p0 = c0 * v0, p1 = c1 * v1;
    x[4 * j4] = p0.x;
    x[4 * j4 + 1] = p0.y;
    x[4 * j4 + 2] = p1.x;
    x[4 * j4 + 3] = p1.y;
  }
  float blIn = 0.f; // BL of this lane's first state
  {
    float blOut = 0.f;
FSMC_Q4_PHASE_LOOP
    for (int ph = 0; ph < 4; ++ph) {
      const float c = quadMove<kQuadDn>(blOut);
      if (qd == ph) {
        blIn = (ph == 0) ? 0.f : c;
        x[0] = blIn + x[0];
#pragma unroll
        for (int j = 1; j < KQ; ++j) {
          x[j] = x[j - 1] + x[j];
        }
        blOut = x[KQ - 1];
      }
    }
  }
  // beta'[k] = (BL[k] + D[k]*vec[k]) + BU[k]
#pragma unroll
  for (int j4 = 0; j4 < (KQ / 4); ++j4) {
    const float4 d = rD[j4];
    const f32x2 v0 = {b[4 * j4], b[4 * j4 + 1]}, v1 = {b[4 * j4 + 2], b[4 * j4 + 3]};
    const f32x2 d0 = {d.x, d.y}, d1 = {d.z, d.w};
    const f32x2 p0 = d0 * v0, p1 = d1 * v1;
    b[4 * j4] = p0.x;
    b[4 * j4 + 1] = p0.y;
    b[4 * j4 + 2] = p1.x;
    b[4 * j4 + 3] = p1.y;
  }
#pragma unroll
  for (int j = 0; j < KQ; ++j) {
    const float bl = (j == 0) ? blIn : x[j - 1];
    b[j] = bl + b[j];
  }
#pragma unroll
  for (int j4 = 0; j4 < (KQ / 4); ++j4) {
    const float4 m = mask[j4]; // 1 for a state of the model, 0 for a ghost
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int j = 4 * j4 + 2 * h;
      const f32x2 t = {b[j], b[j + 1]};
      const f32x2 bu = {w[j], w[j + 1]};
      const f32x2 mm = {f4at(m, 2 * h), f4at(m, 2 * h + 1)};
      const f32x2 r = (t + bu) * mm;
      w[j] = r.x;
      w[j + 1] = r.y;
    }
  }
  const float total = quadOrderedSum(w, qd);
  const float c = 1.0f / total;
  const f32x2 cc = {c, c};
#pragma unroll
  for (int j = 0; j < KQ; j += 2) {
    const f32x2 t = {w[j], w[j + 1]};
    const f32x2 r = t * cc;
    b[j] = r.x;
    b[j + 1] = r.y;
  }
}

// One forward step (HMM.cpp:799-830) + scaling.  a: alpha of site pos-1 on entry, of site pos on exit.
template <int KQ>
__device__ __forceinline__ void alpha_step_q4(float (&a)[KQ], float (&w)[KQ], float (&x)[KQ], const float4* e,
                                              const float4* rD, const float4* rB, const float4* rU, const float4* rC,
                                              const int qd)
{
  // suffix sums one slot down: w[j] = alphaC of the state after j (quarter 3 first)
  {
    float cOut = 0.f; // alphaC of this lane's first state
FSMC_Q4_PHASE_LOOP
    for (int ph = 3; ph >= 0; --ph) {
      const float c = quadMove<kQuadUp>(cOut);
      if (qd == ph) {
        if (ph == 3) {
          w[KQ - 1] = 0.f; // no state after the last one (never used)
          w[KQ - 2] = a[KQ - 1];
        } else {
          w[KQ - 1] = c;
          w[KQ - 2] = w[KQ - 1] + a[KQ - 1];
        }
#pragma unroll
        for (int j = KQ - 2; j >= 1; --j) {
          w[j - 1] = w[j] + a[j];
        }
        cOut = w[0] + a[0];
      }
    }
  }
  // ua = U*alpha (in x), then AU as a running recurrence: x[j] = AU of state j+1
#pragma unroll
  for (int j4 = 0; j4 < (KQ / 4); ++j4) {
    const float4 u = rU[j4];
    const f32x2 v0 = {a[4 * j4], a[4 * j4 + 1]}, v1 = {a[4 * j4 + 2], a[4 * j4 + 3]};
    const f32x2 u0 = {u.x, u.y}, u1 = {u.z, u.w};
    const f32x2 p0 = u0 * v0, p1 = u1 * v1;
    x[4 * j4] = p0.x;
    x[4 * j4 + 1] = p0.y;
    x[4 * j4 + 2] = p1.x;
    x[4 * j4 + 3] = p1.y;
  }
  float auIn = 0.f; // AU of this lane's first state
  {
    float4 crv[KQ / 4]; // this lane's columnRatios, read once ahead of the chain
#pragma unroll
    for (int j4 = 0; j4 < (KQ / 4); ++j4) {
      crv[j4] = rC[j4];
    }
    float auOut = 0.f;
FSMC_Q4_PHASE_LOOP
    for (int ph = 0; ph < 4; ++ph) {
      const float c = quadMove<kQuadDn>(auOut);
      if (qd == ph) {
        auIn = (ph == 0) ? 0.f : c;
        float au = auIn;
#pragma unroll
        for (int j4 = 0; j4 < (KQ / 4); ++j4) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int j = 4 * j4 + i;
            au = x[j] + f4at(crv[j4], i) * au;
            x[j] = au;
          }
        }
        auOut = au;
      }
    }
  }
  // term = AU + D*alpha (+ B*alphaC of the next state), alpha' = e * term
#pragma unroll
  for (int j4 = 0; j4 < (KQ / 4); ++j4) {
    const float4 d = rD[j4];
    const float4 bt = rB[j4];
    const float4 em = e[j4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int j = 4 * j4 + 2 * h;
      const f32x2 av = {a[j], a[j + 1]};
      const f32x2 dd = {f4at(d, 2 * h), f4at(d, 2 * h + 1)};
      const f32x2 bb = {f4at(bt, 2 * h), f4at(bt, 2 * h + 1)};
      const f32x2 ee = {f4at(em, 2 * h), f4at(em, 2 * h + 1)};
      const f32x2 ac = {w[j], w[j + 1]};
      const f32x2 da = dd * av;
      const f32x2 bw = bb * ac;
      f32x2 t;
      t.x = ((j == 0) ? auIn : x[j - 1]) + da.x;
      t.y = x[j] + da.y;
      // (the model's last state has no B term, HMM.cpp:823-826: its alphaC slot is 0, and term + B*0 is term)
      const f32x2 t2 = t + bw;
      const f32x2 o = ee * t2;
      w[j] = o.x;
      w[j + 1] = o.y;
    }
  }
  const float total = quadOrderedSum(w, qd);
  const float c = 1.0f / total;
  const f32x2 cc = {c, c};
#pragma unroll
  for (int j = 0; j < KQ; j += 2) {
    const f32x2 t = {w[j], w[j + 1]};
    const f32x2 r = t * cc;
    a[j] = r.x;
    a[j + 1] = r.y;
  }
}

// Segment ages from the per-state sums of the four lanes of a pair (columns lane0 .. lane0+3 of the wave's plane),
// walked in state order by the pair's first lane (HMM.cpp:1087-1107).
template <int KQ> __device__ __forceinline__ float spsAtQ4(const float4* col0, const int k)
{
  const int qq = k / KQ, j = k - qq * KQ;
  return reinterpret_cast<const float*>(col0 + (size_t)(j >> 2) * kWave + qq)[j & 3];
}
template <int KQ>
__device__ __forceinline__ void segment_ages_q4(const int K, const unsigned nAge, const float4* col0, cfloat_p pi,
                                                cfloat_p expT, const bool wantMean, const bool wantMap, float& mean,
                                                float& mapv)
{
  mean = 0.f;
  mapv = 0.f;
  const int n = (unsigned)K < nAge ? K : (int)nAge;
  if (wantMean) {
    float acc = 0.f;
#pragma nounroll
    for (int k = 0; k < n; ++k) {
      acc = acc + spsAtQ4<KQ>(col0, k);
    }
    const float norm = 1.f / acc;
#pragma nounroll
    for (int k = 0; k < n; ++k) {
      mean = mean + (norm * spsAtQ4<KQ>(col0, k)) * expT[k];
    }
  }
  if (wantMap) {
    float best = 0.f;
    float bestT = 0.f;
#pragma nounroll
    for (int k = 0; k < n; ++k) {
      const float r = spsAtQ4<KQ>(col0, k) / pi[k];
      if (k == 0 || best < r) {
        best = r;
        bestT = expT[k];
      }
    }
    mapv = bestT;
  }
}

// Work item = a quarter of a group: pairs [16*sub, 16*sub+16) of group g; item index = 4*g + sub.
template <int KQ, int MODE, bool TRACK>
__global__ __launch_bounds__(kWave, FSMC_Q4_MINBLOCKS(KQ)) void decode_kernel_q4(const KParams p)
{
  static_assert(MODE == kModeIbd || MODE == kModeDump || MODE == kModePerPair,
                "the wide-model kernel has the IBD, the dump and the per-pair consumer");
  static_assert(KQ % 4 == 0 && KQ <= kQ4MaxStates, "states per lane");
  __shared__ float4 betaLds[(KQ / 4) * kWave]; // landing zone of the next site's beta row (LDS-DMA), 16 KiB at KQ = 64
  __shared__ float4 emisLds[2][3 * KQ];        // two sites x three observation classes (KQ float4 = 4*KQ states a row)
  __shared__ float4 rowLds[2][4 * KQ];         // two sites x four table rows
  __shared__ float4 maskLds[KQ];               // 1.0f for the model's states, 0.0f for the ghosts
  const int K = p.K;                           // states of the model, <= 4*KQ = p.KP

  const int lane = threadIdx.x;
  const int qd = lane & 3; // which quarter of the states
  const int pp = lane >> 2; // which pair of the sub-group
  if (lane < KQ) {
    maskLds[lane] = make_float4(4 * lane < K ? 1.f : 0.f, 4 * lane + 1 < K ? 1.f : 0.f, 4 * lane + 2 < K ? 1.f : 0.f,
                                4 * lane + 3 < K ? 1.f : 0.f);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  waitLgkm0();
  __builtin_amdgcn_wave_barrier();
  const float4* const maskOf = &maskLds[(KQ / 4) * qd];
  const cfloat_p tPi = (cfloat_p)p.pi, tExpT = (cfloat_p)p.expT;
  const cint_p tStepRow = (cint_p)p.stepRow;
  const size_t vecF4 = (size_t)(KQ / 4) * kWave;
  float4* const chunkbuf = p.ws + (size_t)blockIdx.x * p.wsSlot;
  float4* const ckpt = chunkbuf + (size_t)p.chunkRows * vecF4;
  float4* const saveA = ckpt + (size_t)(p.maxChunks + 2) * vecF4;
  float4* const saveS = saveA + vecF4;
  float4* const spsMem = saveS + lane;
  const unsigned laneOff = (unsigned)lane * (unsigned)sizeof(float4); // this lane's byte offset inside a 1-KiB row
  const int C = p.chunk;
  const float4* const rowSets4 = reinterpret_cast<const float4*>(p.rowSets);
  const float4* const cR4 = reinterpret_cast<const float4*>(p.cR);
  const float4* const pi4 = reinterpret_cast<const float4*>(p.pi);

  // LDS-DMA of one site's rows into ring slot (site & 1): asynchronous, no VGPRs.  A slot is only rewritten after
  // every LDS read of the step that used it has returned (s_waitcnt lgkmcnt(0) before the request).
  auto stageEmis = [&](const int site) {
    const float4* src = p.emis3 + (size_t)site * (3 * KQ);
#pragma unroll
    for (int i = 0; i < (3 * KQ + kWave - 1) / kWave; ++i) {
      if (i * kWave + lane < 3 * KQ) {
        dmaToLds((gf32x4_p)(src + i * kWave + lane), &emisLds[site & 1][i * kWave]); // (see dmaToLds: ring slots)
      }
    }
  };
  // rows of the step into / out of `site`: D, B and (forward) U, columnRatios or (backward) Ush, RR -- 4*KQ float4
  // in LDS order; lane idx fetches float4 idx % KQ of row idx / KQ
  auto stageRows = [&](const int site, const bool forward) {
    const float4* rs = rowSets4 + (size_t)tStepRow[site] * (kRowSetParts * KQ);
    const float4* r2 = forward ? rs + kRowU * KQ : rs + kRowUsh * KQ;
    const float4* r3 = forward ? cR4 : rs + kRowRR * KQ;
#pragma unroll
    for (int i = 0; i < (4 * KQ + kWave - 1) / kWave; ++i) {
      const int idx = i * kWave + lane;
      if (idx < 4 * KQ) {
        const int r = idx / KQ;
        const int off = idx - r * KQ;
        const float4* src = (r == 0 ? rs + kRowD * KQ : r == 1 ? rs + kRowB * KQ : r == 2 ? r2 : r3) + off;
        dmaToLds((gf32x4_p)src, &rowLds[site & 1][i * kWave]);
      }
    }
  };
  // (the waits are the s_waitcnt builtin: the compiler's wait-count pass sees them and adds none of its own for the
  //  LDS reads of DMA-landed data -- as inline asm it could not, and guarded every such read with a vmcnt(0))
  auto landed = [&]() { // every outstanding DMA (and store) of this wave is done and visible to its LDS reads
    waitVm0();
    __builtin_amdgcn_wave_barrier();
  };
  // the same while the N youngest vector-memory operations may still be in flight (they retire in order, so "at most
  // N outstanding" means every older request is done): N = the KQ/4 row stores / beta-row DMAs issued AFTER the
  // requests waited for, or the staging requests of the next site issued after the beta row
  constexpr unsigned kRowOps = KQ / 4;                                            // stores or DMAs of one stored row
  constexpr unsigned kStageOps = (3 * KQ + kWave - 1) / kWave + (4 * KQ + kWave - 1) / kWave; // stageEmis + stageRows
  auto landedExceptYoungest = [&](auto n) {
    constexpr unsigned N = decltype(n)::value;
    static_assert(N < 64, "vmcnt is a six-bit counter");
    __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15u) | ((N >> 4) << 14));
    __builtin_amdgcn_wave_barrier();
  };
  auto landedBeforeRowStores = [&]() { landedExceptYoungest(std::integral_constant<unsigned, kRowOps>{}); };
  auto ldsReadsDone = [&]() {
    waitLgkm0();
    __builtin_amdgcn_wave_barrier();
  };
  auto fetchBeta = [&](const float4* row) { // row: wave-uniform address of the stored vector (see store_vec)
    const gchar_p base = uniformPtr(row);
#pragma unroll
    for (int k4 = 0; k4 < (KQ / 4); ++k4) {
#if defined(__HIP_DEVICE_COMPILE__)
      __builtin_amdgcn_global_load_lds(rowSlot(base, k4, laneOff), &betaLds[k4 * kWave], 16, 0, 2 /* nt */);
#endif
    }
  };

  for (;;) {
    unsigned item = 0;
    if (lane == 0) {
      item = atomicAdd(&p.counters[p.groupBase], 1u);
    }
    item = __builtin_amdgcn_readfirstlane(item);
    const unsigned g = item >> 2;
    const int sub = (int)(item & 3u);
    if (g >= (unsigned)p.nGroups) {
      break;
    }
    const cuint_p gw = (cuint_p)(p.groups + g);
    const unsigned firstPair = gw[0];
    const int nPairsInGroup = (int)gw[1];
    const int from = (int)gw[2];
    const int to = (int)gw[3];
    const int scanFrom = (int)gw[4];
    const int aEnd = (MODE == kModeIbd) ? (int)gw[5] : to;
    if (16 * sub >= nPairsInGroup) {
      continue; // this quarter of the group holds no pair
    }
    const int pairInGroup = 16 * sub + pp;
    const bool valid = pairInGroup < nPairsInGroup;
    const unsigned pairIdx = firstPair + (unsigned)(valid ? pairInGroup : 16 * sub);
    const fsmc_pair pr = p.pairs[pairIdx];
    const unsigned long long* rowA = p.haps + (size_t)pr.hap_a * p.W;
    const unsigned long long* rowB = p.haps + (size_t)pr.hap_b * p.W;

    const int nA = aEnd - from;
    const int nChunks = (nA + C - 1) / C;
    const bool single = nChunks <= 1;

    int wordIdx = -1;
    unsigned long long xw = 0, aw = 0;
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
      const int xo = (int)((xw >> bit) & 1ull);
      const int t = (int)((aw >> bit) & 1ull);
      return xo ? 0 : 1 + t;
    };
    // this lane's slices of a staged site
    auto emisOf = [&](const int site, const int cls) -> const float4* {
      return &emisLds[site & 1][cls * KQ + (KQ / 4) * qd];
    };
    auto rowOf = [&](const int site, const int t) -> const float4* {
      return &rowLds[site & 1][t * KQ + (KQ / 4) * qd];
    };

    float w[KQ], x[KQ];

    // one backward step out of site q (rows of q must be staged): beta(q) -> beta(q-1)
    auto betaStep = [&](float (&b)[KQ], const int q) {
      const int c = obsClass(q);
      beta_step_q4<KQ>(b, w, x, emisOf(q, c), rowOf(q, 0), rowOf(q, 1), rowOf(q, 2), rowOf(q, 3), maskOf, qd);
    };
    auto betaInit = [&](float (&b)[KQ]) {
      // all ones, scaled: the sum of K ones is exact (HMM.cpp:887-897); ghosts are zero
      const float c = 1.0f / (float)K;
#pragma unroll
      for (int j4 = 0; j4 < (KQ / 4); ++j4) {
        const float4 m = maskOf[j4];
        b[4 * j4] = m.x * c;
        b[4 * j4 + 1] = m.y * c;
        b[4 * j4 + 2] = m.z * c;
        b[4 * j4 + 3] = m.w * c;
      }
    };

    // ------------------------------------------------------------------ pass B
    {
      float b[KQ];
      betaInit(b);
      auto afterBeta = [&](const int pos) {
        if (single) {
          if (pos < aEnd) {
            store_vec<KQ, KQ>(KQ, chunkbuf + (size_t)(pos - from) * vecF4, laneOff, b);
          }
        } else {
          const int rel = pos - from;
          if (rel > 0 && pos <= aEnd && (rel % C == 0 || pos == aEnd)) {
            const int j = (pos == aEnd) ? nChunks : rel / C;
            store_vec<KQ, KQ>(KQ, ckpt + (size_t)j * vecF4, laneOff, b);
          }
        }
      };
      afterBeta(to - 1);
      if (to - 2 >= from) {
        ldsReadsDone();
        stageEmis(to - 1);
        stageRows(to - 1, false);
      }
      for (int pos = to - 2; pos >= from; --pos) {
        const int q = pos + 1;
        landed();
        if (pos - 1 >= from) {
          stageEmis(q - 1);
          stageRows(q - 1, false);
        }
        betaStep(b, q);
        ldsReadsDone();
        afterBeta(pos);
      }
    }

    // ------------------------------------------------------------------ pass A
    int cur = 4;
    int segStart = 0;
    float acc = 0.f;
    float a[KQ];

    auto emit = [&](const int s0, const int s1) {
      const unsigned idx = atomicAdd(&p.counters[1], 1u);
      float mean = 0.f, mapv = 0.f;
      if constexpr (TRACK) {
        segment_ages_q4<KQ>(K, p.ageThr, spsMem, tPi, tExpT, (p.flags & FSMC_WANT_MEAN) != 0,
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

    for (int jc = 0; jc < (nChunks > 0 ? nChunks : 0); ++jc) {
      const int lo = from + jc * C;
      const int hi = (lo + C < aEnd) ? lo + C : aEnd;
      if (!single) {
        if (jc > 0) {
          store_vec<KQ, KQ>(KQ, saveA, laneOff, a);
        }
        {
          float b[KQ];
          int pos;
          if (hi == to) {
            betaInit(b);
            store_vec<KQ, KQ>(KQ, chunkbuf + (size_t)(to - 1 - lo) * vecF4, laneOff, b);
            pos = to - 2;
          } else {
            load_vec<KQ, KQ>(KQ, ckpt + (size_t)(jc + 1) * vecF4, laneOff, b);
            pos = hi - 1;
          }
          if (pos >= lo) {
            ldsReadsDone();
            stageEmis(pos + 1);
            stageRows(pos + 1, false);
          }
          bool storesBehind = false; // the previous iteration's row stores were issued after this site's requests
          for (; pos >= lo; --pos) {
            const int q = pos + 1;
            if (storesBehind) {
              landedBeforeRowStores();
            } else {
              landed();
            }
            storesBehind = true;
            if (pos - 1 >= lo) {
              stageEmis(q - 1);
              stageRows(q - 1, false);
            }
            betaStep(b, q);
            ldsReadsDone();
            store_vec<KQ, KQ>(KQ, chunkbuf + (size_t)(pos - lo) * vecF4, laneOff, b);
          }
        }
        if (jc > 0) {
          load_vec<KQ, KQ>(KQ, saveA, laneOff, a);
        }
      }

      // the wave's own stores of this chunk's betas must have landed before the DMA reads them back
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      landed();
      ldsReadsDone();
      // Requests run one site ahead of their use, each covered by a whole forward step: the rows of site pos+1 are
      // asked for when site pos opens (their ring slot was last read by site pos-1), the beta row of pos+1 after
      // the combine of pos (one landing zone).  Vector-memory operations retire in order, so the waits count what
      // may stay in flight: the beta row (youngest when a site opens), the next site's rows (youngest at the combine).
      stageEmis(lo);
      stageRows(lo, true); // (unused at the window's first site; keeps the number of requests per site fixed)
      fetchBeta(chunkbuf);
      for (int pos = lo; pos < hi; ++pos) {
        landedExceptYoungest(std::integral_constant<unsigned, kRowOps>{}); // rows of this site
        const bool stagedNext = pos + 1 < hi;
        if (stagedNext) {
          stageEmis(pos + 1);
          stageRows(pos + 1, true);
        }
        const int c = obsClass(pos);
        const float4* e = emisOf(pos, c);
        if (pos == from) {
          // alpha at the first site: pi * emission, scaled (HMM.cpp:736-747)
#pragma unroll
          for (int j4 = 0; j4 < (KQ / 4); ++j4) {
            const float4 pv = pi4[(KQ / 4) * qd + j4];
            const float4 em = e[j4];
            w[4 * j4] = pv.x * em.x;
            w[4 * j4 + 1] = pv.y * em.y;
            w[4 * j4 + 2] = pv.z * em.z;
            w[4 * j4 + 3] = pv.w * em.w;
          }
          const float tot = quadOrderedSum(w, qd);
          const float c0 = 1.0f / tot;
#pragma unroll
          for (int j = 0; j < KQ; ++j) {
            a[j] = w[j] * c0;
          }
        } else {
          alpha_step_q4<KQ>(a, w, x, e, rowOf(pos, 0), rowOf(pos, 1), rowOf(pos, 2), rowOf(pos, 3), qd);
        }
        // combine with beta of this site (landed in LDS) and normalise (HMM.cpp:672-691)
        if (stagedNext) {
          landedExceptYoungest(std::integral_constant<unsigned, kStageOps>{}); // this site's beta row
        } else {
          landed();
        }
#pragma unroll
        for (int j4 = 0; j4 < (KQ / 4); ++j4) {
          const float4 bv = betaLds[j4 * kWave + lane];
          w[4 * j4] = a[4 * j4] * bv.x;
          w[4 * j4 + 1] = a[4 * j4 + 1] * bv.y;
          w[4 * j4 + 2] = a[4 * j4 + 2] * bv.z;
          w[4 * j4 + 3] = a[4 * j4 + 3] * bv.w;
        }
        const float sumq = quadOrderedSum(w, qd);
        const float cq = 1.0f / sumq;
        // every LDS read of this site's rows and of the landing zone has returned: request the next site's
        ldsReadsDone();
        if (pos + 1 < hi) {
          fetchBeta(chunkbuf + (size_t)(pos + 1 - lo) * vecF4);
        }

        if (MODE == kModeDump) {
          float* out = p.dumpOut + p.dumpOffsets[g] + (size_t)(pos - from) * K * kWave + (size_t)(KQ * qd) * kWave +
                       pairInGroup;
          if (valid) {
#pragma unroll
            for (int j = 0; j < KQ; ++j) {
              if (KQ * qd + j < K) {
                out[(size_t)j * kWave] = w[j] * cq;
              }
            }
          }
        }

        if (MODE == kModePerPair) {
          // HMM::writePerPairOutput (HMM.cpp:1378-1409): mean = sum_k post*E[t_k] (k ascending from 0.f), MAP = first
          // strictly larger posterior -- both walk the states in order, quarter after quarter
          const float4* const coal4 = reinterpret_cast<const float4*>(p.expCoal) + (KQ / 4) * qd;
          float mOut = 0.f, bOut = 0.f;
          int aOut = 0;
#pragma nounroll
          for (int ph = 0; ph < 4; ++ph) {
            const float cM = quadMove<kQuadDn>(mOut);
            const float cB = quadMove<kQuadDn>(bOut);
            const int cA = __float_as_int(quadMove<kQuadDn>(__int_as_float(aOut)));
            if (qd == ph) {
              float mean = (ph == 0) ? 0.f : cM;
              float best = (ph == 0) ? 0.f : cB;
              int arg = (ph == 0) ? 0 : cA;
#pragma unroll
              for (int j4 = 0; j4 < (KQ / 4); ++j4) {
                const float4 tc = coal4[j4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                  const float post = w[4 * j4 + i] * cq;
                  mean = mean + post * f4at(tc, i);
                  if (best < post) {
                    arg = KQ * ph + 4 * j4 + i;
                    best = post;
                  }
                }
              }
              mOut = mean;
              bOut = best;
              aOut = arg;
            }
          }
          const float mean = quadMove<kQuadB3>(mOut);
          const int arg = __float_as_int(quadMove<kQuadB3>(__int_as_float(aOut)));
          if (valid && qd == 0) {
            if (p.ppMean) p.ppMean[(size_t)pairIdx * p.S + pos] = mean;
            if (p.ppMap) p.ppMap[(size_t)pairIdx * p.S + pos] = arg;
          }
        }

        if (MODE == kModeIbd) {
          if (pos >= scanFrom) {
            // sum of the posterior over the states below the threshold, in state order across the quarters; the
            // scan normalises whole blocks of four states (as the one-lane kernel does)
            float sOut = 0.f;
#pragma nounroll
            for (int ph = 0; ph < 4; ++ph) {
              const float cIn = quadMove<kQuadDn>(sOut);
              if ((unsigned)(KQ * ph) < p.stateThr) {
                if (qd == ph) {
                  float s = (ph == 0) ? 0.f : cIn;
#pragma unroll
                  for (int j4 = 0; j4 < (KQ / 4); ++j4) {
                    if ((unsigned)(KQ * ph + 4 * j4) < p.stateThr) {
#pragma unroll
                      for (int i = 0; i < 4; ++i) {
                        w[4 * j4 + i] = w[4 * j4 + i] * cq;
                      }
#pragma unroll
                      for (int i = 0; i < 4; ++i) {
                        if ((unsigned)(KQ * ph + 4 * j4 + i) < p.stateThr) s = s + w[4 * j4 + i];
                      }
                    }
                  }
                  sOut = s;
                }
              } else if (qd == ph) {
                sOut = cIn;
              }
            }
            const float s = quadMove<kQuadB3>(sOut);
            const int level = s >= p.thr[0] ? 0 : s >= p.thr[1] ? 1 : s >= p.thr[2] ? 2 : s >= p.thr[3] ? 3 : 4;
            const bool owner = valid && qd == 0; // one lane of the pair writes its records
            if (owner && cur != 4 && level != cur) {
              emit(segStart, pos - 1);
            }
            const bool opening = level != 4 && level != cur;
            if constexpr (TRACK) {
              if (level != 4) {
#pragma unroll
                for (int j4 = 0; j4 < (KQ / 4); ++j4) {
                  if ((unsigned)(KQ * qd + 4 * j4) < p.ageThr) {
                    float4 sv = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (!opening) {
                      sv = spsMem[(size_t)j4 * kWave];
                    }
                    const float sc = ((unsigned)(KQ * qd + 4 * j4) < p.stateThr) ? 1.0f : cq;
                    sv.x = sv.x + w[4 * j4] * sc;
                    sv.y = sv.y + w[4 * j4 + 1] * sc;
                    sv.z = sv.z + w[4 * j4 + 2] * sc;
                    sv.w = sv.w + w[4 * j4 + 3] * sc;
                    spsMem[(size_t)j4 * kWave] = sv;
                  }
                }
              }
            }
            acc = (level == 4) ? 0.f : (opening ? s : acc + s);
            if (opening) {
              segStart = pos;
            }
            cur = level;
            if (pos == aEnd - 1) {
              if (owner && cur != 4) {
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
