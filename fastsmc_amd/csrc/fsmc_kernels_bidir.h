// fsmc_kernels_bidir.h -- two waves per decode window for the consumers that have no state across sites (sums over
// pairs, per-pair mean / MAP rows, posterior dump), for launches that leave at least half the chip's wave slots empty.
//
// decode_kernel (fsmc_kernels.h) gives a group of <= 64 pairs ONE wave: a beta sweep down the window, then the alpha
// sweep up, 2 steps per site one after the other.  A launch of fewer groups than SIMDs (the FASTSMC_EXAMPLE shape: 701
// groups on 1024 SIMDs; ASMC.decodePairs lists of a few thousand pairs) leaves every such wave alone on its SIMD, where it
// issues at most every fourth cycle, while the other SIMDs idle.  The posterior of a site is alpha * beta: the alpha
// recursion starts at the window's first site, the beta recursion at its last -- two independent chains.  Here a
// workgroup of TWO waves decodes one group, lane = pair in both:
//   phase 1   wave A: alpha of sites from .. mid-1, every row stored  |  wave B: beta of sites to-1 .. mid, every row stored
//   barrier   (both waves' rows have reached memory)
//   phase 2   wave A: alpha of sites mid .. to-1, combined with B's stored beta rows -> consumer
//             wave B: beta of sites mid-1 .. from, combined with A's stored alpha rows -> consumer
// Every site gets one alpha step and one beta step -- the steps of the single-wave kernel, the same functions
// (alpha_step_pk, beta_step_pk, HMM.cpp:799-830, 957-1016), the same operands in the same order -- and one combine
// (HMM.cpp:672-691: alpha[k] * beta[k], the sum over k ascending, 1.0f / sum), so every posterior is the same bits; a
// window takes each wave half the steps.  The consumers (HMM.cpp:1044-1085, 1378-1409) do not care in which order the
// sites arrive.  The IBD scan does (HMM.cpp:1179-1357: a state machine over ascending sites) and stays with one wave.
// Array mode only; the whole window's rows live in the workgroup's workspace slot (to - from rows), so the host uses
// this kernel when the plan keeps windows whole and the launch has at most half as many groups as the chip has slots.
#pragma once

#include "fsmc_kernels.h"

namespace fsmc
{

// Workgroup barrier for the two role loops (each wave executes the same number of them, from its own copy of the loop):
// LDS traffic of this wave has returned; vector memory is waited for by the callers where it matters.
__device__ __forceinline__ void bidirBarrier()
{
  FSMC_GCN_ASM("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int KT, int MODE>
__global__ __launch_bounds__(2 * kWave, minWavesPerSimd(KT)) void decode_kernel_bidir(const KParams p)
{
  static_assert(MODE == kModeDump || MODE == kModeSums || MODE == kModePerPair, "the consumers without state across sites");
  static_assert(KT > 0 && KT <= 128, "a member of the lane-per-pair family");
  constexpr int KA = KT;
  constexpr int K = KT;
  constexpr int K4A = (KA + 3) / 4;
  constexpr int K4 = K4A;
  constexpr int E4A = ((KA + kKPad - 1) / kKPad) * (kKPad / 4);
  constexpr int NC = 3;
  constexpr int NL = (NC * E4A + kWave - 1) / kWave;
  const int Kreal = kGhost<KT> ? p.K : K;
  const int KP = p.KP;
  const int E4 = KP >> 2;

  // per wave: the two-site emission ring and the landing zone of the other wave's stored row (the sums consumer also
  // transposes the K x 64 posterior tile through it, row stride 65 floats: fsmc_kernels.h)
  constexpr int kLandF4 = (MODE == kModeSums && (KA * 65 + 3) / 4 > K4A * kWave) ? (KA * 65 + 3) / 4 : K4A * kWave;
  __shared__ float4 emisLdsAll[2][2][NC * E4A];
  __shared__ float4 landLdsAll[2][kLandF4];
  __shared__ unsigned groupLds;
  // kModePerPair: the expected coalescence times, read from LDS in the consumer's loop over the states.  (As scalar loads --
  // one per state, all live at once: hundreds of spilled scalars -- the 128-state member's instantiation of this kernel
  // returned wrong and, with several groups, irreproducible means; with the times in LDS it is bit-equal like the others.)
  __shared__ float coalLds[MODE == kModePerPair ? KA : 1];

  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // 0: wave A (forward), 1: wave B (backward)
  if (MODE == kModePerPair) {
    for (int k = threadIdx.x; k < K; k += 2 * kWave) {
      coalLds[k] = (k < (kGhost<KT> ? p.K : K)) ? p.expCoal[k] : 0.f;
    }
    __syncthreads();
  }
  // This wave's LDS is always addressed as emisLdsAll[wave][...] / landLdsAll[wave][...], straight off the __shared__
  // arrays: through a pointer VARIABLE (float4* land = landLdsAll[wave]) the accesses go through a generic-to-LDS address
  // cast whose null check this compiler folds wrongly in some instantiations -- one member's requests landed at wrong LDS
  // addresses (its sums were off), another did not assemble ("operand has incorrect register class").
  // the ring's LDS byte address for the asm requests: the low half of its generic address (the aperture's offset) -- no
  // generic-to-LDS cast, whose null check this compiler folds into an instruction it then rejects in some members
  const unsigned emisLdsAddr =
      (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)(const void*)&emisLdsAll[wave][0][0]);
  const cfloat_p tPi = (cfloat_p)p.pi, tCR = (cfloat_p)p.cR;
  const Tables tabs = {(cfloat_p)p.rowSets, tCR, (cfloat_p)p.ghostMask};
  const size_t vecF4 = (size_t)K4 * kWave;
  float4* const rows = p.ws + (size_t)blockIdx.x * p.wsSlot; // [to - from] stored vectors: alpha below mid, beta from mid on
  const unsigned laneOff = (unsigned)lane * (unsigned)sizeof(float4);

  // The two roles are two instantiations of one generic lambda, each with its own copy of the group loop: the roles' live
  // ranges never meet (one body with `if (wave == 0) ... else ...` inside every phase made the register allocator juggle
  // both roles' values at once: hundreds of spilled scalars at the wide members).  Both copies execute the same number of
  // workgroup barriers per group.
  auto runRole = [&](auto roleTag) __attribute__((always_inline)) {
  constexpr int wv = decltype(roleTag)::value; // 0: wave A (forward), 1: wave B (backward)
  for (unsigned round = 0;; ++round) {
    unsigned g = 0;
    if (MODE == kModeSums) { // one BATCH per workgroup and launch, its groups in turn (fsmc_kernels.h, same place)
      if (p.batchFirst) {
        const cuint_p bf = (cuint_p)p.batchFirst;
        g = bf[p.groupBase + blockIdx.x] + round;
        if (g >= bf[p.groupBase + blockIdx.x + 1]) {
          g = (unsigned)p.nGroups;
        }
      } else {
        g = round == 0 ? (unsigned)p.groupBase + blockIdx.x : (unsigned)p.nGroups;
      }
    } else {
      if (wv == 0 && lane == 0) {
        groupLds = atomicAdd(&p.counters[p.groupBase], 1u);
      }
      bidirBarrier();
      g = __builtin_amdgcn_readfirstlane(groupLds);
      bidirBarrier(); // (rewritten by the next round only after both waves have read it)
    }
    if (g >= (unsigned)p.nGroups) {
      break;
    }
    const cuint_p gw = (cuint_p)(p.groups + (size_t)g);
    const unsigned firstPair = gw[0];
    const int nPairsInGroup = (int)gw[1];
    const int from = (int)gw[2];
    const int to = (int)gw[3];
    const int mid = from + (to - from) / 2; // wave A stores alpha of [from, mid), wave B beta of [mid, to)
    const bool valid = lane < nPairsInGroup;
    const unsigned pairIdx = firstPair + (valid ? (unsigned)lane : 0u);
    const fsmc_pair pr = p.pairs[pairIdx];
    const unsigned long long* rowA = p.haps + (size_t)pr.hap_a * p.W;
    const unsigned long long* rowB = p.haps + (size_t)pr.hap_b * p.W;

    int wordIdx = -1;
    unsigned long long xw = 0, aw = 0;
    auto obsClass = [&](const int q) -> int { // 0 het, 1 hom major, 2 hom minor (HMM.cpp:647-652)
      const int wi = q >> 6;
      if (__builtin_expect(wi != wordIdx, 0)) {
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
    auto stageEmis = [&](const int q) { // a site's three emission rows into ring slot q & 1, by LDS-DMA (NL requests)
      const gchar_p src = uniformPtr(p.emis3 + (size_t)q * (NC * E4));
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const int idx = lane + i * kWave;
        if (idx < NC * E4) {
          // (the slot's LDS address as a value the compiler KNOWS to be wave-uniform: under register pressure it otherwise
          //  hands the asm statement's scalar operand a vector register -- "operand has incorrect register class")
          const unsigned slotAddr = emisLdsAddr + (unsigned)(((q & 1) * (NC * E4A) + i * kWave) * sizeof(float4));
#if defined(__HIP_DEVICE_COMPILE__)
          // M0 is saved and restored around the request: the compiler does not honour an "m0" clobber (it warns that the
          // register is reserved) and may keep the landing zone's address of its own LDS-DMA requests in M0 across
          // this statement -- in some instantiations it did, and the next row then landed in the emission ring
          unsigned keepM0;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                       : "=&s"(keepM0)
                       : "v"((gf32x4_p)(src + (size_t)i * (kWave * sizeof(float4)) + laneOff)), "s"(slotAddr)
                       : "memory");
#endif
        }
      }
    };
    int rowBlk = -1;
    int rowVec = 0;
    auto stepRowOf = [&](const int site) -> int { // table row of the step into `site` (fsmc_kernels.h, same place)
      const int blk = site >> 6;
      if (__builtin_expect(blk != rowBlk, 0)) {
        const int idx = blk * kWave + lane;
        rowVec = p.stepRow[idx < p.S ? idx : p.S - 1];
        rowBlk = blk;
        waitVm0();
      }
      return __builtin_amdgcn_readlane(rowVec, site & (kWave - 1));
    };
    // "at most n vector-memory operations outstanding" (they retire in order): the older emission-row request is done
    auto waitVmAtMost = [&](auto nTag) {
#if defined(FSMC_BIDIR_FULL_WAITS) // (diagnostic: every counted wait becomes a full one)
      constexpr unsigned n = 64;
#else
      constexpr unsigned n = decltype(nTag)::value;
#endif
      if constexpr (n < 64) {
        __builtin_amdgcn_s_waitcnt(0x0F70 | (n & 15u) | ((n >> 4) << 14));
      } else {
        waitVm0();
      }
    };
    // the other wave's stored vector of one site into this wave's landing zone (LDS-DMA, K4 requests)
    auto fetchRow = [&](const float4* row) {
      const gchar_p base = uniformPtr(row);
#pragma unroll
      for (int k4 = 0; k4 < K4; ++k4) {
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_global_load_lds(rowSlot(base, k4, laneOff), &landLdsAll[wave][k4 * kWave], 16, 0, 2 /* nt */);
#endif
      }
    };

    float w[KA];
    float v[KA]; // wave A: alpha, wave B: beta
    Diag dg;

    // w = v * (the landed vector), sumq = its sum over the states, k ascending from 0.f (HMM.cpp:672-691)
    auto combine = [&]() -> float {
      float sumq = 0.f;
      constexpr int kCB = 8;
      constexpr int NB = (K + kCB - 1) / kCB;
      auto loadB = [&](const int blk, float4& b0, float4& b1) {
        b0 = landLdsAll[wave][(2 * blk) * kWave + lane];
        b1 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (2 * blk + 1 < K4) {
          b1 = landLdsAll[wave][(2 * blk + 1) * kWave + lane];
        }
      };
      float4 c0, c1;
      loadB(0, c0, c1);
#pragma unroll
      for (int blk = 0; blk < NB; ++blk) {
        float4 n0 = c0, n1 = c1;
        if (blk + 1 < NB) {
          loadB(blk + 1, n0, n1);
        }
#pragma unroll
        for (int i = 0; i < kCB; i += 2) {
          const int k = blk * kCB + i;
          if (k + 1 < K) {
            const f32x2 av = {v[k], v[k + 1]};
            const f32x2 bv = {pick(c0, c1, i), pick(c0, c1, i + 1)};
            const f32x2 q = pmul(av, bv);
            w[k] = q.x;
            w[k + 1] = q.y;
            sumq = sumq + q.x;
            sumq = sumq + q.y;
          } else {
#pragma unroll
            for (int ii = i; ii < i + 2; ++ii) {
              const int kk = blk * kCB + ii;
              if (kk < K) {
                w[kk] = v[kk] * pick(c0, c1, ii);
                sumq = sumq + w[kk];
              }
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        c0 = n0;
        c1 = n1;
      }
      return sumq;
    };

    // The consumers of one site's posterior w * cq.  `c`: this lane's observation class at the site; `freeSlot`: the ring
    // slot whose rows are no longer needed (the sums' 00 / 01 / 11 split parks the classes of the 64 pairs there).
    auto consume = [&](const int pos, const float cq, const int c, const int freeSlot) {
      if (MODE == kModePerPair) { // HMM.cpp:1378-1409
        float mean = 0.f;
        float best = 0.f;
        int arg = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const float post = w[k] * cq;
          mean = mean + post * coalLds[k];
          if (best < post) {
            arg = k;
            best = post;
          }
        }
        if (valid) {
          if (p.ppMean) p.ppMean[(size_t)pairIdx * p.S + pos] = mean;
          if (p.ppMap) p.ppMap[(size_t)pairIdx * p.S + pos] = arg;
        }
      }
      if (MODE == kModeDump) {
        float* out = p.dumpOut + p.dumpOffsets[g] + (size_t)(pos - from) * Kreal * kWave + lane;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          if (!kGhost<KT> || k < Kreal) {
            out[(size_t)k * kWave] = valid ? w[k] * cq : 0.f;
          }
        }
      }
      if (MODE == kModeSums) {
        // HMM::augmentSumOverPairs (HMM.cpp:1052-1081), as in fsmc_kernels.h: the K x 64 tile transposed through the
        // landing zone, lane j owns state j (and j + 64), the pairs of the batch added in batch order
        float* const tile = reinterpret_cast<float*>(&landLdsAll[wave][0]);
        unsigned char* const cls = reinterpret_cast<unsigned char*>(&emisLdsAll[wave][freeSlot][0]);
#pragma unroll
        for (int k = 0; k < K; ++k) {
          tile[k * 65 + lane] = w[k] * cq;
        }
        if (p.flags & FSMC_WANT_MAJOR_MINOR_SUMS) {
          cls[lane] = (unsigned char)c;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int kb = 0; kb < Kreal; kb += 2 * kWave) {
          const int kk0 = kb + lane, kk1 = kb + kWave + lane;
          const bool h0 = kk0 < Kreal, h1 = kk1 < Kreal;
          float* const acc0 = p.sums + (size_t)blockIdx.x * p.sumsSlot + (size_t)pos * Kreal + (h0 ? kk0 : 0);
          float* const acc1 = p.sums + (size_t)blockIdx.x * p.sumsSlot + (size_t)pos * Kreal + (h1 ? kk1 : 0);
          float s[2] = {0.f, 0.f}, s00[2] = {0.f, 0.f}, s01[2] = {0.f, 0.f}, s11[2] = {0.f, 0.f};
          if (round > 0) { // a later group of the batch: the running sums of the pairs before (THIS wave wrote them: the
                           // two waves split every group of the batch at the same site)
            if (p.flags & FSMC_WANT_SUMS) {
              if (h0) s[0] = acc0[0];
              if (h1) s[1] = acc1[0];
            }
            if (p.flags & FSMC_WANT_MAJOR_MINOR_SUMS) {
              if (h0) {
                s00[0] = acc0[p.sumsPlane];
                s01[0] = acc0[2 * p.sumsPlane];
                s11[0] = acc0[3 * p.sumsPlane];
              }
              if (h1) {
                s00[1] = acc1[p.sumsPlane];
                s01[1] = acc1[2 * p.sumsPlane];
                s11[1] = acc1[3 * p.sumsPlane];
              }
            }
          }
          const int t0 = (h0 ? kk0 : 0) * 65, t1 = (h1 ? kk1 : 0) * 65;
          auto walk = [&](auto splitTag) {
            constexpr bool SPLIT = decltype(splitTag)::value;
            constexpr int kWalk = 16;
            auto add = [&](const float q0, const float q1, const int cv) {
              s[0] = s[0] + q0;
              s[1] = s[1] + q1;
              if constexpr (SPLIT) { // 0 het -> 01, 1 hom major -> 00, 2 hom minor -> 11
                s11[0] = s11[0] + (cv == 2 ? q0 : 0.f);
                s11[1] = s11[1] + (cv == 2 ? q1 : 0.f);
                s00[0] = s00[0] + (cv == 1 ? q0 : 0.f);
                s00[1] = s00[1] + (cv == 1 ? q1 : 0.f);
                s01[0] = s01[0] + (cv == 0 ? q0 : 0.f);
                s01[1] = s01[1] + (cv == 0 ? q1 : 0.f);
              }
            };
            int vv = 0;
            for (; vv + kWalk <= nPairsInGroup; vv += kWalk) {
              float q0[kWalk], q1[kWalk];
              int cv[kWalk];
#pragma unroll
              for (int i = 0; i < kWalk; ++i) {
                q0[i] = tile[t0 + vv + i];
                q1[i] = tile[t1 + vv + i];
                cv[i] = SPLIT ? (int)cls[vv + i] : 0;
              }
#pragma unroll
              for (int i = 0; i < kWalk; ++i) {
                add(q0[i], q1[i], cv[i]);
              }
            }
            for (; vv < nPairsInGroup; ++vv) {
              add(tile[t0 + vv], tile[t1 + vv], SPLIT ? (int)cls[vv] : 0);
            }
          };
          if (p.flags & FSMC_WANT_MAJOR_MINOR_SUMS) {
            walk(std::true_type{});
          } else {
            walk(std::false_type{});
          }
          if (p.flags & FSMC_WANT_SUMS) {
            if (h0) acc0[0] = s[0];
            if (h1) acc1[0] = s[1];
          }
          if (p.flags & FSMC_WANT_MAJOR_MINOR_SUMS) {
            if (h0) {
              acc0[p.sumsPlane] = s00[0];
              acc0[2 * p.sumsPlane] = s01[0];
              acc0[3 * p.sumsPlane] = s11[0];
            }
            if (h1) {
              acc1[p.sumsPlane] = s00[1];
              acc1[2 * p.sumsPlane] = s01[1];
              acc1[3 * p.sumsPlane] = s11[1];
            }
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        waitLgkm0();
        __builtin_amdgcn_wave_barrier();
      }
    };

    if constexpr (wv == 0) {
      // ---------------------------------------------------------------- wave A: alpha, ascending
      // ring: slot pos & 1 holds the rows of site pos; the rows of site pos + 2 are requested once the step into pos is
      // over.  Per iteration the wave issues K4 row stores, then NL ring requests: at the top of the next iteration the
      // rows it needs were requested before the last K4 + NL operations.
      stageEmis(from);
      if (from + 1 < to) {
        stageEmis(from + 1);
      }
      waitVm0();
      for (int pos = from; pos < mid; ++pos) {
        if (pos > from) {
          waitVmAtMost(std::integral_constant<unsigned, (unsigned)(K4 + NL)>{});
        }
        __builtin_amdgcn_wave_barrier();
        const int c = obsClass(pos);
        const float4* e = &emisLdsAll[wave][pos & 1][c * E4];
        if (__builtin_expect(pos == from, 0)) {
          alpha_init<KT, KA>(K, v, tPi, e);
        } else {
          alpha_step_pk<KT, KA>(v, w, rowSetOf<KT>(tabs, stepRowOf(pos)), tCR, e, dg);
        }
        store_vec<KT, KA>(K, rows + (size_t)(pos - from) * vecF4, laneOff, v);
        if (pos + 2 < to) {
          stageEmis(pos + 2);
        }
      }
    } else {
      // ---------------------------------------------------------------- wave B: beta, descending
      // beta at the window's last site (HMM.cpp:887-897), then the step out of site q = pos + 1 gives beta of site pos;
      // ring: slot q & 1 holds the rows of site q, the rows of q - 1 are requested at the top of the iteration into the
      // other slot (whose step is over)
      beta_init<KT, KA>(K, p.K, v);
      if (to - 2 >= from) {
        stageEmis(to - 1); // (requested BEFORE the row store: "at most K4 outstanding" then means it has landed)
      }
      if (to - 1 >= mid) {
        store_vec<KT, KA>(K, rows + (size_t)(to - 1 - from) * vecF4, laneOff, v);
      }
      for (int pos = to - 2; pos >= mid; --pos) {
        const int q = pos + 1;
        // the rows of q were requested before the K4 stores of the iteration before (the first time: before the store of
        // the initial vector, if that was stored -- it always is: to - 1 >= mid)
        waitVmAtMost(std::integral_constant<unsigned, (unsigned)K4>{});
        __builtin_amdgcn_wave_barrier();
        if (pos - 1 >= from) {
          stageEmis(q - 1);
        }
        const int c = obsClass(q);
        beta_step_pk<KT, KA, true, kGhost<KT>>(v, w, rowSetOf<KT>(tabs, stepRowOf(q)), &emisLdsAll[wave][q & 1][c * E4],
                                               tabs.ghostMask, dg);
        store_vec<KT, KA>(K, rows + (size_t)(pos - from) * vecF4, laneOff, v);
      }
    }
    // both waves' rows are in memory before either reads the other's
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    waitVm0();
    bidirBarrier();
#if defined(FSMC_BIDIR_ACQUIRE_AGENT) // (diagnostic: invalidate this CU's vector cache before reading the other wave's rows)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
#if defined(FSMC_BIDIR_SERIAL) // (diagnostic: wave B's second phase only after wave A's)
    if constexpr (wv == 1) {
      bidirBarrier();
    }
#endif
    if constexpr (wv == 0) {
      // ---------------------------------------------------------------- wave A, phase 2: sites mid .. to-1
      if (mid < to) {
        fetchRow(rows + (size_t)(mid - from) * vecF4);
      }
      for (int pos = mid; pos < to; ++pos) {
        const int c = obsClass(pos);
        const float4* e = &emisLdsAll[wave][pos & 1][c * E4];
        // everything requested so far has landed: this site's emission rows (requested two sites ago) and beta row
        // (requested behind the combine of the site before)
        waitVm0();
        __builtin_amdgcn_wave_barrier();
        if (__builtin_expect(pos == from, 0)) {
          alpha_init<KT, KA>(K, v, tPi, e);
        } else {
          alpha_step_pk<KT, KA>(v, w, rowSetOf<KT>(tabs, stepRowOf(pos)), tCR, e, dg);
        }
        const float sumq = combine();
        const float cq = 1.0f / sumq;
        waitLgkm0(); // every read of the landing zone and of this site's ring slot has returned
        if (MODE != kModeSums) {
          if (pos + 1 < to) {
            fetchRow(rows + (size_t)(pos + 1 - from) * vecF4);
          }
          if (pos + 2 < to) {
            stageEmis(pos + 2);
          }
        }
        consume(pos, cq, c, pos & 1);
        if (MODE == kModeSums) { // (the tile went through the landing zone, the classes through the ring slot)
          if (pos + 1 < to) {
            fetchRow(rows + (size_t)(pos + 1 - from) * vecF4);
          }
          if (pos + 2 < to) {
            stageEmis(pos + 2);
          }
        }
      }
    } else {
      // ---------------------------------------------------------------- wave B, phase 2: sites mid-1 .. from
      if (mid - 1 >= from) {
        fetchRow(rows + (size_t)(mid - 1 - from) * vecF4);
      }
      for (int pos = mid - 1; pos >= from; --pos) {
        const int q = pos + 1;
        waitVm0(); // the rows of site q (requested an iteration ago) and alpha of site pos have landed
        __builtin_amdgcn_wave_barrier();
        const int c = obsClass(q);
        beta_step_pk<KT, KA, true, kGhost<KT>>(v, w, rowSetOf<KT>(tabs, stepRowOf(q)), &emisLdsAll[wave][q & 1][c * E4],
                                               tabs.ghostMask, dg);
        const int cPos = obsClass(pos); // (the class of THIS site: what the 00 / 01 / 11 split of the sums asks for)
        const float sumq = combine();
        const float cq = 1.0f / sumq;
        waitLgkm0();
        if (MODE != kModeSums) {
          if (pos - 1 >= from) {
            fetchRow(rows + (size_t)(pos - 1 - from) * vecF4);
            stageEmis(pos); // the rows of the next step's site q' = pos, into the slot of site q + 1
          }
        }
        consume(pos, cq, cPos, q & 1); // (slot q & 1: the step out of q is over)
        if (MODE == kModeSums) {
          if (pos - 1 >= from) {
            fetchRow(rows + (size_t)(pos - 1 - from) * vecF4);
            stageEmis(pos);
          }
        }
      }
    }
#if defined(FSMC_BIDIR_SERIAL)
    if constexpr (wv == 0) {
      waitVm0();
      bidirBarrier();
    }
#endif
    // the next group reuses the rows and the queue word: both waves are through with this one
    waitVm0();
    bidirBarrier();
  }
  };
  if (wave == 0) {
    runRole(std::integral_constant<int, 0>{});
  } else {
    runRole(std::integral_constant<int, 1>{});
  }
}

} // namespace fsmc
