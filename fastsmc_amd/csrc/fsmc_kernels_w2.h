// fsmc_kernels_w2.h -- the decode kernel for wide models (128 < K <= 1024; K = 256 is BASELINE.json config 4) with
// lane = pair and SEVERAL WAVES per group (what follows is written for four waves of up to 64 states, the members of up to
// 256 states; the members beyond: w2Member, fsmc_capi.hip).
//
// A lane cannot hold the K-vectors of a 256-state model (2 x 256 registers is the whole file), and splitting a pair over
// four lanes (the kernel this one replaced) leaves three quarters of a wave idle in every recurrence.  Here a workgroup of NW = 4 waves
// decodes one group of <= 64 pairs: lane l of EVERY wave is pair l, wave h holds states [KH*h, KH*h + KH) (KH = 48 or 64)
// -- every instruction of every wave serves 64 pairs, and a wave's two K-vectors are 2 x 64 registers (two waves per
// SIMD).  The first-order recurrences cross the boundaries between the waves through a mailbox in LDS
// and workgroup barriers; they come in opposite pairs, which pipeline against each other over NW phases:
//   backward step   phase p: wave NW-1-p runs BU (descending) over its states | wave p runs BL (ascending) over its states
//   forward step    phase p: wave NW-1-p runs the suffix sums alphaC (desc.)   | wave p runs AU (ascending)
// (a wave of the lower half runs its ascending pass first, one of the upper half its descending pass), then the scaling
// sum walks the waves in state order.  A wave works in two of the NW phases; the other workgroup on the CU (two fit:
// their beta landing zones fill LDS) runs in the gaps.  Every value is produced by the same IEEE operation on the same
// operands, in the same order, as in the reference (HMM.cpp:799-830, 957-1016, HmmUtils.cpp:102-151): bit-identical.
// Operands are wave-uniform (each wave its own part of the table rows): scalar loads one block ahead, as in
// fsmc_kernels.h; each wave stages the emission values of its own states in its own two-site LDS ring and lands its own
// part of the next beta row by LDS-DMA.  Beta stride 1; array mode and sequence mode; consumers: IBD scan (with segment
// ages), posterior dump, sums over pairs, per-pair mean / MAP rows.  Ghost padding: rows padded to KP = NW*KH floats with
// zero table, emission and prior entries -- ghost values stay exactly +0 through every operation, except that beta' of a
// ghost is BL: ghosts only occur in the upper half, whose backward step multiplies by the 1/0 mask row (x * 1.0f is
// exact).
#pragma once

#include <type_traits>

#include "fsmc_kernels.h"

namespace fsmc
{

constexpr int kW2NW = 4;    // waves per group of the members of up to 256 (and 257 ... 320) states; NW = 6 ... 8 beyond
constexpr int kW2MaxNW = 8;
// mailbox rows (64 floats each): carries of the recurrences per boundary between two waves, partial sums per wave
template <int NW> struct W2Rows {
  static constexpr int T = 0, BU = NW - 1, BL = 2 * (NW - 1), C = 0, AU = NW - 1; // (+ boundary 0 .. NW-2)
  static constexpr int Step = 3 * (NW - 1), Comb = Step + NW, Level = Comb + NW;   // (+ wave; Level: two rows in turn)
  static constexpr int Scan = Step; // (the scan's partial sums follow the combine: the step's rows are free then)
  static constexpr int Mean = T;    // (kModePerPair: likewise the carries' rows)
  static constexpr int Mail = Level + 2;
};
constexpr int kWBF = 8;    // ... and of the forward pass (four tables at a time)
constexpr int kWBWide = 8;  // ... of the passes with two operand rows (16-state blocks measured 4 % slower at 64 states per wave: spills)
constexpr int kWB = 8;     // states per operand block of the backward passes here (two waves' roles in one kernel leave
                           // fewer scalar registers than fsmc_kernels.h has: 16-state blocks were spilled in flight)

// Workgroup barrier that publishes this wave's LDS writes and nothing else: no vmcnt wait (the beta-row stores of the
// step before are still on their way to HBM).  The "memory" clobber keeps the compiler's LDS accesses on their side.
__device__ __forceinline__ void w2Barrier()
{
  FSMC_GCN_ASM("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// The barriers between the phases of a step / between the partial sums.  -DFSMC_W2_DIAG_NO_PHASE_BARRIERS and
// -DFSMC_W2_DIAG_NO_SUM_BARRIERS are timing experiments only (results are wrong): what the hand-overs cost.
__device__ __forceinline__ void w2PhaseBarrier()
{
#if !defined(FSMC_W2_DIAG_NO_PHASE_BARRIERS)
  w2Barrier();
#endif
}
__device__ __forceinline__ void w2SumBarrier()
{
#if !defined(FSMC_W2_DIAG_NO_SUM_BARRIERS)
  w2Barrier();
#endif
}

// A stored K-vector's traffic rides inside the step functions, a piece (one float4 per lane, 1 KiB per wave) at a time
// between the operand blocks of a wave's first pass, instead of going out as a burst of sixteen between two steps.  One
// CU moves about 10 B per clock to and from HBM -- a hundred cycles per piece -- and a wave that issues a burst stands at
// the issue of every piece until the queue has room (region stamps: 12 % of the kernel in the burst behind the combine,
// 8-10 % around the row stores); spread over a pass the pieces find the queue empty.
//   out: the row the step STARTS from (beta of site q, final since the end of the step before) goes to HBM piece by piece
//        just before each block of it is overwritten;
//   in:  the next site's beta row is requested into the wave's landing zone (LDS-DMA) during the forward step, whose
//        combine has released the zone.
// Whether a step moves a row is a compile-time parameter of the step functions (IO): a wave-uniform branch around every
// piece split the operand blocks' straight-line code and cost hundreds of spilled scalars.  A step of an IO
// instantiation that has no row to move is given a spare row of the workspace instead.
struct RowIO {
  gchar_p base;      // wave-uniform address of this wave's part of the row (lane 0, first state)
  unsigned laneOff;  // 16 * lane
  float4* lds;       // in: this wave's landing zone
};
__device__ __forceinline__ RowIO noRowIO()
{
  const RowIO r = {nullptr, 0u, nullptr};
  return r;
}
// pieces [first, first + n) of the row the step starts from, out of registers
template <int KH>
__device__ __forceinline__ void rowOutPieces(const RowIO& io, const float (&v)[KH], const int first, const int n)
{
#pragma unroll
  for (int k4 = first; k4 < first + n; ++k4) {
    const f32x4 ov = {v[4 * k4], v[4 * k4 + 1], v[4 * k4 + 2], v[4 * k4 + 3]};
    __builtin_nontemporal_store(ov, rowSlot(io.base, k4, io.laneOff));
  }
}
// pieces [first, first + n) of the next row, into the landing zone
__device__ __forceinline__ void rowInPieces(const RowIO& io, const int first, const int n)
{
#pragma unroll
  for (int k4 = first; k4 < first + n; ++k4) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_global_load_lds(rowSlot(io.base, k4, io.laneOff), &io.lds[k4 * kWave], 16, 0, 2 /* nt */);
#endif
  }
}

// Scalar-cache warm-up in the idle phases.  A pass waits for its operand blocks one at a time (scalar loads return out
// of order: the only usable wait is lgkmcnt(0)), and a block that misses the 16-KB scalar cache costs an L2 round trip --
// several hundred cycles while the beta stream keeps L2 busy, against ~130 cycles of arithmetic per block.  A wave works
// in two of a step's four phases; in a phase it would spend at the barrier it requests one dword of every 64-byte line of
// the rows its coming passes read (its own part of them), and the phase's barrier waits for them: those passes' blocks then
// hit.  (The outer waves' first pass is phase 0: nothing to hide behind.  Warming it during the sum of the step before
// was measured: the requests outlast the sum and the step got longer.)
template <int KH, int KP, int R0, int R1 = -1, int R2 = -1, int R3 = -1> struct WarmRows {
  static constexpr int kLines = KH / 16 < 4 ? KH / 16 : 4; // (a Touched holds four lines: the first four of a wider part)
  Touched t0, t1, t2, t3;
  __device__ __forceinline__ void request(cfloat_p rsw)
  {
    touchRow<0, kLines>(t0, rsw, R0 * KP);
    if constexpr (R1 >= 0) touchRow<0, kLines>(t1, rsw, R1 * KP);
    if constexpr (R2 >= 0) touchRow<0, kLines>(t2, rsw, R2 * KP);
    if constexpr (R3 >= 0) touchRow<0, kLines>(t3, rsw, R3 * KP);
  }
  __device__ __forceinline__ void landed() const // behind the phase's barrier (it waits for lgkmcnt(0))
  {
    heldRow<kLines>(t0);
    if constexpr (R1 >= 0) heldRow<kLines>(t1);
    if constexpr (R2 >= 0) heldRow<kLines>(t2);
    if constexpr (R3 >= 0) heldRow<kLines>(t3);
  }
};

// The lambdas that take a K-vector by reference are inlined whatever their size: left to its cost model the inliner keeps
// the site step of an eight-wave member (eight roles' code in one body) as a function, and the K-vectors -- and the
// kernel's parameters, captured by reference -- then live in scratch memory.
#define FSMC_W2_INLINE __attribute__((always_inline))

struct W2Ctx {
  float* mail;  // [W2Rows<NW>::Mail][64] in LDS, shared by the waves of the group
  int lane;
  int h;        // which wave: states [KH*h, KH*h + KH)
  bool hi;      // a wave of the upper half: descending pass first
};

// Sum of all states in state order (HmmUtils.cpp:121-128): wave 0 adds its states from 0.f, every next wave continues.
// NW barriers; returns the total in every wave.  `row`: first of NW mailbox rows.
// FIRST = 1: wave 0's partial sum is in the mailbox already, published by the barrier the caller has just passed (the
// step functions: wave 0's row is final at the end of the step's last phase, so it adds its states before that
// phase's barrier instead of behind it -- one hand-over less per step).
template <int KH, int H> __device__ __forceinline__ float w2PartialSum(const float (&v)[KH], float s)
{
#pragma unroll
  for (int k = 0; k < KH; ++k) {
    s = s + v[k];
  }
  return s;
}
template <int NW, int KH, int H, int FIRST = 0>
__device__ __forceinline__ float w2OrderedTotal(const W2Ctx& cx, const float (&v)[KH], const int row)
{
#pragma unroll
  for (int ph = FIRST; ph < NW; ++ph) {
    if (H == ph) {
      const float s = w2PartialSum<KH, H>(v, ph == 0 ? 0.f : cx.mail[(row + ph - 1) * kWave + cx.lane]);
      cx.mail[(row + ph) * kWave + cx.lane] = s;
    }
    if (ph == NW - 1) {
      w2Barrier();
    } else {
      w2SumBarrier();
    }
  }
  return cx.mail[(row + NW - 1) * kWave + cx.lane];
}
#if defined(FSMC_W2_NO_MERGED_SUM) // (A/B switch: the separate first hand-over of round 3)
constexpr int kW2SumFirst = 0;
#else
constexpr int kW2SumFirst = 1;
#endif

// v = w * (1.0f / total) (HmmUtils.cpp:102-151)
template <int KH> __device__ __forceinline__ void w2Scale(float (&v)[KH], const float (&w)[KH], const float total)
{
  const float c = 1.0f / total;
  const f32x2 cc = {c, c};
#pragma unroll
  for (int k = 0; k < KH; k += 2) {
    const f32x2 x = {w[k], w[k + 1]};
    const f32x2 y = pmul(x, cc);
    v[k] = y.x;
    v[k + 1] = y.y;
  }
}

// One backward step (HMM.cpp:957-1016).  b: this wave's half of beta of site pos+1 on entry, of site pos on exit.
// rs: the step's RowSet (all 2*KH states); e: this lane's emission values of THIS WAVE's states (LDS).
template <int NW, int KH, int H, bool SCALE = true, bool IO = false>
__device__ __forceinline__ void beta_step_w2(const W2Ctx& cx, float (&b)[KH], float (&w)[KH], cfloat_p rs,
                                             const float4* e, cfloat_p ghostMask, Diag& dg, const RowIO& out)
{
  FSMC_END(dg, 20); // (region stamps, diagnostic builds: 0-3 work of the phases, 4-7 their barriers, 8 sum, 9 scale)
  constexpr int KP = NW * KH;
  // 64-byte lines of this wave's part of a table row that the warm-up touches (a Touched holds four; the members of
  // more than 64 states a wave warm the first lines of their part only)
  constexpr int kLines = KH / 16 < 4 ? KH / 16 : 4;
  static_assert(KH % kWBWide == 0 && KH % 16 == 0, "whole operand blocks and lines");
  long long dummy = 0;
  (void)dummy;
  constexpr int off = H * KH;  // first state of this wave
  const cfloat_p rsw = rs + off;  // this wave's states of the RowSet rows (block offsets stay instruction immediates)
  const cfloat_p gmw = ghostMask + off;
  // ---- descending pass: vec[k] = beta[k]*e[k] (kept in b), T[k] = Ush[k]*vec[k], BU[k] = T[k+1] + RR[k]*BU[k+1]
  // wave 1 runs it in phase 0 (BU above the last state is 0), wave 0 in phase 1 from wave 1's (T, BU) of state KH
  auto descending = [&](auto blockStates, const float tIn, const float buIn, const bool accumulate) {
    constexpr int BS = decltype(blockStates)::value; // states per operand block of this pass
    constexpr int NB = KH / BS;
    typedef typename SV<BS>::T SVec;
    // accumulate = false: w[k] = BU[k];  true: w[k] = w[k] + BU[k] (w holds BL + D*vec already)
    SVec u, rr, nu, nrr;
    EmisBlk<BS> em, nem;
    u = LD<BS, false>::loadAt(rsw, kRowUsh * KP + (NB - 1) * BS);
    rr = LD<BS, false>::loadAt(rsw, kRowRR * KP + (NB - 1) * BS);
    // scalar-cache warm-up (fsmc_kernels.h, touchLines): one dword of every other 64-byte line of this wave's part of the
    // two rows, so that the pass's first wait covers all the misses at once and the later blocks hit
    Touched tu, trr;
    touchRow<0, kLines - 1>(tu, rsw, kRowUsh * KP);
    touchRow<0, kLines - 1>(trr, rsw, kRowRR * KP);
    if (!accumulate) {
      em = readEmis<BS>(e, NB - 1);
    }
    float tAbove = tIn;  // T of the state above the current one
    float buAbove = buIn; // BU of the state above the current one
#pragma unroll
    for (int blk = NB - 1; blk >= 0; --blk) {
      FSMC_WAIT_OPERANDS(dummy);
      if (blk == NB - 1) {
        landed(u, rr);
        heldRow<kLines - 1>(tu);
        heldRow<kLines - 1>(trr);
      } else {
        landed(nu, nrr);
        u = nu;
        rr = nrr;
        if (!accumulate) {
          em = nem;
        }
      }
      if (blk > 0) {
        nu = LD<BS, false>::loadAt(rsw, kRowUsh * KP + (blk - 1) * BS);
        nrr = LD<BS, false>::loadAt(rsw, kRowRR * KP + (blk - 1) * BS);
        if (!accumulate) {
          nem = readEmis<BS>(e, blk - 1);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (IO && H == NW - 1) {
        if (!accumulate) {
          rowOutPieces<KH>(out, b, blk * (BS / 4), BS / 4); // (this block of the row is about to become vec)
        }
      }
      float T[BS];
#pragma unroll
      for (int i = 0; i < BS; i += 2) {
        const int k = blk * BS + i;
        f32x2 v = {b[k], b[k + 1]};
        if (!accumulate) {
          v = pmul(v, em.pair(i)); // (in phase 1 wave 0's b already holds vec: the ascending pass made it)
          b[k] = v.x;
          b[k + 1] = v.y;
        }
        const f32x2 t = pmul(pairOf(u, i), v);
        T[i] = t.x;
        T[i + 1] = t.y;
      }
      // the recurrence (two dependent operations a state), then -- second pass of a lower-half wave -- the block's BU
      // added to what the ascending pass left in w, two states an instruction (as a plain add a state inside the chain
      // loop it was 64 instructions of the 287 on this pass's way, and this pass is the longer one of its phase)
      float BUb[BS];
#pragma unroll
      for (int i = BS - 1; i >= 0; --i) {
        const float tNext = (i == BS - 1) ? tAbove : T[i + 1];
        const float bu = tNext + rr[i] * buAbove;
        buAbove = bu;
        BUb[i] = bu;
      }
#pragma unroll
      for (int i = 0; i < BS; i += 2) {
        const int k = blk * BS + i;
        f32x2 x = {BUb[i], BUb[i + 1]};
        if (accumulate) {
          const f32x2 wv = {w[k], w[k + 1]};
          x = padd(wv, x);
        }
        w[k] = x.x;
        w[k + 1] = x.y;
      }
      tAbove = T[0];
    }
    if (H > 0) { // carry for the states below: (T, BU) of this wave's first state
      cx.mail[(W2Rows<NW>::T + H - 1) * kWave + cx.lane] = tAbove;
      cx.mail[(W2Rows<NW>::BU + H - 1) * kWave + cx.lane] = buAbove;
    }
  };
  // ---- ascending pass: BL[k] = BL[k-1] + B[k-1]*vec[k-1];  x[k] = BL[k] + D[k]*vec[k]
  // wave 0 runs it in phase 0 (BL[0] = 0; it also forms vec), wave 1 in phase 1 from wave 0's BL of state KH
  // Ghost states (K ... KP-1) lie in the LAST wave only -- a model is given the member whose waves it fills all but the
  // last of -- except at 48 states a wave (K = 129 ... 192: from 129 to 143 states the third wave has ghosts too): only
  // those waves multiply their row by the 1/0 mask (and load it).
  // (An instantiation WITHOUT any mask for models that fill every wave -- config 4's 256 = 4 x 64 states -- was built and
  //  measured in round 5: 3 % SLOWER at size and 8 % on 3000-site windows, interleaved on one box: the allocator's luck,
  //  not the instruction count, decides this kernel.  Not kept.)
  // (the members of more than 512 states -- eight waves of 80 / 96 / 128 -- serve every model that fills all but their last
  //  TWO waves: 513 ... 560 states leave the seventh of eight 80-state waves with ghosts too.  The same for eight waves of
  //  64, to serve 385 ... 448 states in place of the seven-wave member -- an odd group, slower than eight waves: 645
  //  against 612 ms -- was measured: the second masked wave costs the eight-wave member exactly that, 612 -> 645 ms.)
  constexpr bool kMasked = KH == 48 ? (2 * H >= NW) : (KP > 512 ? H >= NW - 2 : H == NW - 1);
  auto ascending = [&](auto blockStates, const float blIn, const bool first) {
    constexpr int BS = decltype(blockStates)::value;
    constexpr int NB = KH / BS;
    typedef typename SV<BS>::T SVec;
    // first = true (wave 0): b = beta on entry, vec on exit; w[k] = x[k].  false (wave 1): w[k] = (x[k] + BU[k]) * mask
    SVec d, bt, mk, nd, nbt, nmk;
    EmisBlk<BS> em, nem;
    d = LD<BS, false>::loadAt(rsw, kRowD * KP);
    bt = LD<BS, false>::loadAt(rsw, kRowB * KP);
    Touched td, tbt;
    touchRow<1, kLines - 1>(td, rsw, kRowD * KP);
    touchRow<1, kLines - 1>(tbt, rsw, kRowB * KP);
    if (first) {
      em = readEmis<BS>(e, 0);
    } else if constexpr (kMasked) {
      mk = LD<BS, false>::loadAt(gmw, 0);
    }
    float BL = blIn;
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
      FSMC_WAIT_OPERANDS(dummy);
      if (blk > 0) {
        landed(nd, nbt);
        d = nd;
        bt = nbt;
        if (first) {
          em = nem;
        } else if constexpr (kMasked) {
          landed(nmk);
          mk = nmk;
        }
      } else {
        landed(d, bt);
        heldRow<kLines - 1>(td);
        heldRow<kLines - 1>(tbt);
        if constexpr (kMasked) {
          if (!first) {
            landed(mk);
          }
        }
      }
      if (blk + 1 < NB) {
        nd = LD<BS, false>::loadAt(rsw, kRowD * KP + (blk + 1) * BS);
        nbt = LD<BS, false>::loadAt(rsw, kRowB * KP + (blk + 1) * BS);
        if (first) {
          nem = readEmis<BS>(e, blk + 1);
        } else if constexpr (kMasked) {
          nmk = LD<BS, false>::loadAt(gmw, (blk + 1) * BS);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (IO && H == 0) {
        if (first) {
          rowOutPieces<KH>(out, b, blk * (BS / 4), BS / 4); // (this block of the row is about to become vec)
        }
      }
      // upper half: beta' of a ghost state is BL, not 0 -- every block is multiplied by its part of the 1/0 mask row
      // (x * 1.0f is exact; ghosts only occur in the upper half)
#pragma unroll
      for (int i = 0; i < BS; i += 2) {
        const int k = blk * BS + i;
        f32x2 v = {b[k], b[k + 1]};
        if (first) {
          v = pmul(v, em.pair(i));
          b[k] = v.x;
          b[k + 1] = v.y;
        }
        const f32x2 dv = pmul(pairOf(d, i), v);
        const f32x2 bv = pmul(pairOf(bt, i), v);
        f32x2 bl;
        bl.x = BL;
        bl.y = BL + bv.x;
        f32x2 x = padd(bl, dv);
        if (!first) {
          const f32x2 bu = {w[k], w[k + 1]};
          x = padd(x, bu);
          if constexpr (kMasked) {
            x = pmul(x, pairOf(mk, i));
          }
        }
        w[k] = x.x;
        w[k + 1] = x.y;
        BL = bl.y + bv.y;
      }
    }
    if (H < NW - 1) {
      cx.mail[(W2Rows<NW>::BL + H) * kWave + cx.lane] = BL; // BL of the next wave's first state
    }
  };
#pragma unroll
  for (int ph = 0; ph < NW; ++ph) {
    if (H == ph) { // the ascending pass reaches this wave
      const float blIn = ph == 0 ? 0.f : cx.mail[(W2Rows<NW>::BL + ph - 1) * kWave + cx.lane];
      if ((2 * H >= NW)) {
        ascending(std::integral_constant<int, kWB>{}, blIn, false); // three operand rows: the smaller blocks
      } else {
        ascending(std::integral_constant<int, kWBWide>{}, blIn, true);
      }
    }
    if (H == NW - 1 - ph) { // the descending pass reaches this wave
      const float tIn = ph == 0 ? 0.f : cx.mail[(W2Rows<NW>::T + H) * kWave + cx.lane];
      const float buIn = ph == 0 ? 0.f : cx.mail[(W2Rows<NW>::BU + H) * kWave + cx.lane];
      if ((2 * H >= NW)) {
        descending(std::integral_constant<int, kWBWide>{}, tIn, buIn, false);
      } else {
        descending(std::integral_constant<int, kWBWide>{}, tIn, buIn, true);
      }
    }
    if constexpr (IO && H != 0 && H != NW - 1) {
      // the inner waves' first pass comes in phase 1: their part of the row goes out in phase 0, which they would
      // spend at the barrier (the outer waves' first pass IS phase 0: their pieces go between its operand blocks)
      if (ph == 0) {
        rowOutPieces<KH>(out, b, 0, KH / 4);
      }
    }
    // idle-phase warm-up (WarmRows): the inner waves sit out phase 0 and then read D, B and Ush, RR in phases 1 and 2;
    // the outer waves sit out phase 1 and read the rows of their second pass in phase 3
    WarmRows<KH, KP, kRowD, kRowB, kRowUsh, kRowRR> warmInner;
    WarmRows<KH, KP, (H == 0 ? kRowUsh : kRowD), (H == 0 ? kRowRR : kRowB)> warmOuter;
    constexpr bool outer = H == 0 || H == NW - 1;
    if (!outer && ph == 0) {
      warmInner.request(rsw);
    }
    if (outer && ph == 1) {
      warmOuter.request(rsw);
    }
    if constexpr (SCALE && kW2SumFirst == 1 && H == 0) {
      if (ph == NW - 1) { // wave 0's row is final: its share of the ordered sum rides on this phase's barrier
        cx.mail[W2Rows<NW>::Step * kWave + cx.lane] = w2PartialSum<KH, H>(w, 0.f);
      }
    }
    FSMC_END(dg, ph);
    w2PhaseBarrier();
    if (!outer && ph == 0) {
      warmInner.landed();
    }
    if (outer && ph == 1) {
      warmOuter.landed();
    }
    FSMC_END(dg, 4 + ph);
  }
  if constexpr (SCALE) {
    const float total = w2OrderedTotal<NW, KH, H, kW2SumFirst>(cx, w, W2Rows<NW>::Step);
    FSMC_END(dg, 8);
    w2Scale<KH>(b, w, total);
    FSMC_END(dg, 9);
  } else { // the un-normalised half-step of sequence mode (HMM.cpp:915-922)
#pragma unroll
    for (int k = 0; k < KH; ++k) {
      b[k] = w[k];
    }
  }
}

// One forward step (HMM.cpp:799-830) + scaling.  a: this wave's half of alpha of site pos-1 on entry, of pos on exit.
template <int NW, int KH, int H, bool SCALE = true, bool IO = false>
__device__ __forceinline__ void alpha_step_w2(const W2Ctx& cx, float (&a)[KH], float (&w)[KH], cfloat_p rs, cfloat_p cR,
                                              const float4* e, Diag& dg, const RowIO& in)
{
  FSMC_END(dg, 21); // (10-13 work of the phases, 14-17 their barriers, 18 sum, 19 scale)
  constexpr int KP = NW * KH;
  constexpr int NBF = KH / kWBF;
  constexpr int kLines = KH / 16 < 4 ? KH / 16 : 4;
  static_assert(KH % kWBF == 0 && KH % 16 == 0, "whole operand blocks and lines");
  typedef typename SV<kWBF>::T SVec;
  long long dummy = 0;
  (void)dummy;
  constexpr int off = H * KH;
  const cfloat_p rsw = rs + off;
  const cfloat_p crw = cR + off;
  // ---- AU ascending: AU[k] = U[k-1]*alpha[k-1] + colRatio[k-1]*AU[k-1];  term = (AU + D*alpha) (+ B*alphaC[k+1])
  // wave 0 in phase 0 (AU[0] = 0; the B term follows in phase 1), wave 1 in phase 1 from wave 0's AU of state KH with
  // its suffix sums (w[k] = alphaC[k+1]) at hand: w[k] = e[k]*term
  auto ascending = [&](const float auIn, const bool complete) {
    SVec d, u, c4, nd, nu, nc;
    EmisBlk<kWBF> em, nem;
    d = LD<kWBF, false>::loadAt(rsw, kRowD * KP);
    u = LD<kWBF, false>::loadAt(rsw, kRowU * KP);
    c4 = LD<kWBF, false>::loadAt(crw, 0);
    Touched td, tu;
    touchRow<1, kLines - 1>(td, rsw, kRowD * KP);
    touchRow<1, kLines - 1>(tu, rsw, kRowU * KP);
    if (complete) {
      em = readEmis<kWBF>(e, 0);
    }
    float AU = auIn;
#pragma unroll
    for (int blk = 0; blk < NBF; ++blk) {
      FSMC_WAIT_OPERANDS(dummy);
      if (blk > 0) {
        landed(nd, nu);
        landed(nc);
        d = nd;
        u = nu;
        c4 = nc;
        if (complete) {
          em = nem;
        }
      } else {
        landed(d, u);
        landed(c4);
        heldRow<kLines - 1>(td);
        heldRow<kLines - 1>(tu);
      }
      if (blk + 1 < NBF) {
        nd = LD<kWBF, false>::loadAt(rsw, kRowD * KP + (blk + 1) * kWBF);
        nu = LD<kWBF, false>::loadAt(rsw, kRowU * KP + (blk + 1) * kWBF);
        nc = LD<kWBF, false>::loadAt(crw, (blk + 1) * kWBF);
        if (complete) {
          nem = readEmis<kWBF>(e, blk + 1);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < kWBF; i += 2) {
        const int k = blk * kWBF + i;
        const f32x2 av = {a[k], a[k + 1]};
        const f32x2 da = pmul(pairOf(d, i), av);
        const f32x2 ua = pmul(pairOf(u, i), av);
        f32x2 au;
        au.x = AU;
        au.y = ua.x + c4[i] * AU; // AU of state k+1
        f32x2 term = padd(au, da);
        if (complete) {
          // w holds B[k] * alphaC[k+1] (the suffix pass formed the product on its way down)
          const f32x2 bw = {w[k], w[k + 1]};
          term = padd(term, bw);
          term = pmul(em.pair(i), term);
        }
        w[k] = term.x;
        w[k + 1] = term.y;
        AU = ua.y + c4[i + 1] * au.y; // AU of state k+2
      }
    }
    if (H < NW - 1) {
      cx.mail[(W2Rows<NW>::AU + H) * kWave + cx.lane] = AU; // AU of the next wave's first state
    }
  };
  // ---- suffix sums, descending: alphaC[k] = alphaC[k+1] + alpha[k] (HMM.cpp:799-814)
  // upper half: w[k] = B[k] * alphaC of the state above k (what the term of state k adds, HMM.cpp:823-826: the product is
  // formed here, on the way down, in the phase the lower half's longer pass sets the pace of -- the upper half's
  // ascending pass, the longest of the step, is that much shorter); the model's last state has no state above it: its
  // alphaC is 0 and B*0 = +0 leaves its term unchanged
  auto suffix = [&](const float cIn) {
    SVec bt, nbt;
    bt = LD<kWBF, false>::loadAt(rsw, kRowB * KP + (NBF - 1) * kWBF);
    Touched tb;
    touchRow<0, kLines - 1>(tb, rsw, kRowB * KP);
    float c = cIn;
#pragma unroll
    for (int blk = NBF - 1; blk >= 0; --blk) {
      FSMC_WAIT_OPERANDS(dummy);
      if (blk == NBF - 1) {
        landed(bt);
        heldRow<kLines - 1>(tb);
      } else {
        landed(nbt);
        bt = nbt;
      }
      if (blk > 0) {
        nbt = LD<kWBF, false>::loadAt(rsw, kRowB * KP + (blk - 1) * kWBF);
      }
      __builtin_amdgcn_sched_barrier(0);
      float ac[kWBF];
#pragma unroll
      for (int i = kWBF - 1; i >= 0; --i) {
        ac[i] = c;
        c = c + a[blk * kWBF + i];
      }
#pragma unroll
      for (int i = 0; i < kWBF; i += 2) {
        const int k = blk * kWBF + i;
        const f32x2 acv = {ac[i], ac[i + 1]};
        const f32x2 bw = pmul(pairOf(bt, i), acv);
        w[k] = bw.x;
        w[k + 1] = bw.y;
      }
    }
    cx.mail[(W2Rows<NW>::C + H - 1) * kWave + cx.lane] = c; // alphaC of this wave's first state (H >= 2 here)
  };
  // lower half (w holds AU + D*alpha already): w[k] = e[k]*(w[k] + B[k]*alphaC[k+1]) on the way down
  auto finish = [&](const float cIn) {
    SVec bt, nbt;
    EmisBlk<kWBF> em, nem;
    bt = LD<kWBF, false>::loadAt(rsw, kRowB * KP + (NBF - 1) * kWBF);
    Touched tb;
    touchRow<0, kLines - 1>(tb, rsw, kRowB * KP);
    em = readEmis<kWBF>(e, NBF - 1);
    float c = cIn; // alphaC of the state above the current one
#pragma unroll
    for (int blk = NBF - 1; blk >= 0; --blk) {
      FSMC_WAIT_OPERANDS(dummy);
      if (blk == NBF - 1) {
        landed(bt);
        heldRow<kLines - 1>(tb);
      } else {
        landed(nbt);
        bt = nbt;
        em = nem;
      }
      if (blk > 0) {
        nbt = LD<kWBF, false>::loadAt(rsw, kRowB * KP + (blk - 1) * kWBF);
        nem = readEmis<kWBF>(e, blk - 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      // the suffix sums of the block first (a chain of adds), then the products two states at a time
      float ac[kWBF];
#pragma unroll
      for (int i = kWBF - 1; i >= 0; --i) {
        ac[i] = c;
        c = c + a[blk * kWBF + i];
      }
#pragma unroll
      for (int i = 0; i < kWBF; i += 2) {
        const int k = blk * kWBF + i;
        const f32x2 acv = {ac[i], ac[i + 1]};
        const f32x2 bw = pmul(pairOf(bt, i), acv);
        f32x2 term = {w[k], w[k + 1]};
        term = padd(term, bw);
        term = pmul(em.pair(i), term);
        w[k] = term.x;
        w[k + 1] = term.y;
      }
    }
    if (H > 0) {
      cx.mail[(W2Rows<NW>::C + H - 1) * kWave + cx.lane] = c;
    }
  };
#pragma unroll
  for (int ph = 0; ph < NW; ++ph) {
    if (H == ph) { // the AU recurrence reaches this wave
      const float auIn = ph == 0 ? 0.f : cx.mail[(W2Rows<NW>::AU + ph - 1) * kWave + cx.lane];
      if ((2 * H >= NW)) {
        ascending(auIn, true);
      } else {
        ascending(auIn, false);
      }
    }
    if (H == NW - 1 - ph) { // the suffix sums reach this wave
      const float cIn = ph == 0 ? 0.f : cx.mail[(W2Rows<NW>::C + H) * kWave + cx.lane];
      if ((2 * H >= NW)) {
        suffix(cIn);
      } else {
        finish(cIn);
      }
    }
    if constexpr (IO) {
      // the next beta row is requested in the phases a wave would spend at the barrier: the outer waves work in the first
      // and the last phase, the inner ones in between (four waves: phases 1 and 2) -- half a row in each of two idle
      // phases (the combine is a sum pass away)
      constexpr bool outer = H == 0 || H == NW - 1;
#if defined(FSMC_W2_ROWIN_EARLY) // (experiment: the whole row in a wave's FIRST idle phase)
      if ((outer && ph == 1) || (!outer && ph == 0)) {
        rowInPieces(in, 0, KH / 4);
      }
#else
      if ((outer && ph == 1) || (!outer && ph == 0)) {
        rowInPieces(in, 0, KH / 8);
      }
      if ((outer && ph == 2) || (!outer && ph == NW - 1)) {
        rowInPieces(in, KH / 8, KH / 4 - KH / 8);
      }
#endif
    }
    // idle-phase warm-up (WarmRows): wave 1 reads D, U in phase 1 and B in phase 2, wave 2 B in phase 1 and D, U in phase
    // 2 -- both sit out phase 0; wave 0 reads B and wave 3 D, U in phase 3 -- both sit out phase 1 (the column ratios are
    // one row for every step: always cached)
    WarmRows<KH, KP, kRowD, kRowU, kRowB> warmInner;
    WarmRows<KH, KP, (H == 0 ? kRowB : kRowD), (H == 0 ? -1 : kRowU)> warmOuter;
    constexpr bool outerA = H == 0 || H == NW - 1;
    if (!outerA && ph == 0) {
      warmInner.request(rsw);
    }
    if (outerA && ph == 1) {
      warmOuter.request(rsw);
    }
    if constexpr (SCALE && kW2SumFirst == 1 && H == 0) {
      if (ph == NW - 1) { // (as in the backward step: wave 0's share of the ordered sum rides on this barrier)
        cx.mail[W2Rows<NW>::Step * kWave + cx.lane] = w2PartialSum<KH, H>(w, 0.f);
      }
    }
    FSMC_END(dg, 10 + ph);
    w2PhaseBarrier();
    if (!outerA && ph == 0) {
      warmInner.landed();
    }
    if (outerA && ph == 1) {
      warmOuter.landed();
    }
    FSMC_END(dg, 14 + ph);
  }
  if constexpr (SCALE) {
    const float total = w2OrderedTotal<NW, KH, H, kW2SumFirst>(cx, w, W2Rows<NW>::Step);
    FSMC_END(dg, 18);
    w2Scale<KH>(a, w, total);
    FSMC_END(dg, 19);
  } else { // the un-normalised half-step of sequence mode (HMM.cpp:760-767)
#pragma unroll
    for (int k = 0; k < KH; ++k) {
      a[k] = w[k];
    }
  }
}

// The wave's role is a compile-time parameter of the step functions (its phases are then straight-line code); the
// kernel branches on the wave number once per call.
// (the last wave's role is the switch's default: the compiler need not prove h < NW)
#define FSMC_W2_ROLE_CASE(n, CALL)                                                                                     \
  if constexpr (n < NW - 1) {                                                                                          \
    if (h_ == n) {                                                                                                     \
      constexpr int H = n;                                                                                             \
      CALL;                                                                                                            \
      break;                                                                                                           \
    }                                                                                                                  \
  }
#define FSMC_W2_ROLE(h, CALL)                                                                                          \
  do {                                                                                                                 \
    const int h_ = (h);                                                                                                \
    FSMC_W2_ROLE_CASE(0, CALL)                                                                                         \
    FSMC_W2_ROLE_CASE(1, CALL)                                                                                         \
    FSMC_W2_ROLE_CASE(2, CALL)                                                                                         \
    FSMC_W2_ROLE_CASE(3, CALL)                                                                                         \
    FSMC_W2_ROLE_CASE(4, CALL)                                                                                         \
    FSMC_W2_ROLE_CASE(5, CALL)                                                                                         \
    FSMC_W2_ROLE_CASE(6, CALL)                                                                                         \
    {                                                                                                                  \
      constexpr int H = NW - 1;                                                                                        \
      CALL;                                                                                                            \
    }                                                                                                                  \
  } while (0)

// Work item = one group of <= 64 pairs; workgroup = NW waves; two workgroups per CU (the landing zones fill LDS).
// SEQ: sequence mode (two steps per site, a fourth emission row per site: fsmc_kernels.h) -- the same schedule as there.
// (-DFSMC_W2_WG_PER_CU=1, an experiment: one workgroup per CU gets 512 registers a lane -- 256 + 102 accumulation
//  registers, no scratch instead of 165 spilled -- and runs 1.43 times faster by itself, but the CU then idles in the two
//  phases of four a wave has no work in: C4 3.59 -> 4.96 s.)
#ifndef FSMC_W2_WG_PER_CU
#define FSMC_W2_WG_PER_CU 2
#endif
// The member of 80 states a wave (models of 257 ... 320 states in four waves) runs ONE workgroup per CU: its landing
// zones alone are 80 KiB of the CU's 160, and a wave has the whole 512-entry register file (256 registers +
// accumulation registers for what the allocator has to park).  So do the groups of more than four waves (NW = 6 ... 8
// of 64 states: 321 ... 512 states; landing zones of 96 ... 128 KiB; more than four waves a CU leave a wave 256
// registers, the four-wave 64-state member's budget).
constexpr int w2WorkgroupsPerCU(int KH, int NW = kW2NW) { return NW == kW2NW && KH <= 64 ? FSMC_W2_WG_PER_CU : 1; }
// members built with resident chunks (the kernel, same name)
constexpr bool w2ResidentBuilt(int NW, bool SEQ) { return NW == kW2NW && !SEQ; }
template <int KH, int MODE, bool TRACK, bool SEQ = false, int NW = kW2NW>
__global__ __launch_bounds__(NW * kWave, w2WorkgroupsPerCU(KH, NW)) void decode_kernel_w2(const KParams p)
{
  static_assert(NW >= 4 && NW <= kW2MaxNW, "waves per group");
  static_assert(MODE == kModeIbd || MODE == kModeDump || MODE == kModeSums || MODE == kModePerPair,
                "the consumers of the wave-group kernel");
  constexpr int KP = NW * KH;
  constexpr int K4H = KH / 4;        // float4 per lane of this wave's part of a K-vector
  constexpr int NC = SEQ ? 4 : 3;    // emission rows per site: three observation classes (+ the gap's homozygous row)
  // The rows of the three observation classes lie one float4 further apart than their length (array mode): K4H * 16 B is
  // a multiple of the LDS bank window, so lanes of different classes reading "their" row at the same state index hit
  // the same banks -- a three-way conflict on every emission read (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 30 % in
  // round 3's counters); one float4 of padding per row spreads the classes over different banks.  The padding lane of
  // a row's LDS-DMA request reads the row's last float4 again.  (Sequence mode keeps the dense layout: four padded rows
  // would need a second request per site.)
#if defined(FSMC_W2_NO_EMIS_PAD) // (A/B switch)
  constexpr int ERS = K4H;
#else
  constexpr int ERS = SEQ ? K4H : K4H + 1; // float4 between the rows of two classes in a ring slot
#endif
  constexpr int E4H = NC * ERS;      // float4 of one site's emission values of this wave's states (+ padding)
  constexpr int NLE = (E4H + kWave - 1) / kWave;
  __shared__ float4 emisLds[NW][2][E4H];       // [wave][ring slot][class * K4H + k4]
  // Members of more than 512 states (eight waves of 80 ... 128) have NO landing zones -- a group's beta row alone would be
  // the CU's LDS: the combine reads the row from the workspace into registers instead (behind the forward step, its
  // latency in the open), and the sums consumer transposes its tile 32 states a turn.
  constexpr bool LAND = KP <= 512;
  constexpr int kTileStates = LAND ? KH : 32; // kModeSums: states of a turn through the tile
  constexpr int kLandF4 = LAND ? K4H * kWave : (MODE == kModeSums ? kTileStates * kWave / 4 : 1);
  __shared__ float4 betaLds[NW][kLandF4];      // [wave]: landing zone of the next site's beta row (its part)
  __shared__ float mailLds[W2Rows<NW>::Mail * kWave];
  __shared__ float4 piLds[KP / 4];   // initialStateProb, zero padded
  __shared__ float4 coalLds[MODE == kModePerPair ? KP / 4 : 1]; // kModePerPair: expected coalescence times, zero padded
  __shared__ unsigned groupLds;
  __shared__ unsigned char clsLds[NW][kWave]; // kModeSums: observation class of every pair at the current site
  // Per-lane bookkeeping that lives for a whole group sits in LDS, not in registers: the XOR / AND words of the
  // current 64 sites of every pair and the table rows of the steps into those sites.  (As registers they were the
  // values the allocator spilled in the site loops -- reloaded from scratch memory at every site behind a vmcnt(0),
  // i.e. behind the beta row and the emission rows that had just been requested.)  Lane l of EVERY wave is pair l, so
  // the four waves compute and write the same values to the same words (a benign race: a wave reads a block's words
  // only behind its own write, and only in front of the first barrier of a site's step, while no wave can be a
  // site ahead of another).
  __shared__ unsigned long long obsLds[2][kWave]; // [XOR | AND][pair]
  __shared__ int rowLds[2][kWave];                // [64-site block % 2][site % 64]

  const int lane = threadIdx.x & (kWave - 1);
  // (wave-uniform BY CONSTRUCTION: as a scalar the compiler branches on it instead of predicating both roles)
  const int h = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const W2Ctx cx = {mailLds, lane, h, 2 * h >= NW};
  const int K = p.K; // states of the model, <= KP
  const cfloat_p tPi = (cfloat_p)p.pi, tExpT = (cfloat_p)p.expT;
  const size_t vecF4 = (size_t)(KP / 4) * kWave; // float4 per stored K-vector of the group
  const size_t halfF4 = (size_t)h * K4H * kWave;  // this wave's half inside a stored vector
  float4* const chunkbuf = p.ws + (size_t)blockIdx.x * p.wsSlot;
  float4* const ckpt = chunkbuf + (size_t)p.chunkRows * vecF4;
  float4* const saveA = ckpt + (size_t)(p.maxChunks + 2) * vecF4;
  float4* const saveS = saveA + vecF4;
  float4* const spsMem = saveS + lane; // this lane's column of the per-state posterior sums (all states)
  // Resident chunks (array mode; the host plans p.residentChunks extra chunk buffers per workgroup, fsmc_kernels.h, same
  // place): pass B leaves the rows of the window's FIRST chunks -- the last ones it walks through, the first ones the
  // alpha sweep needs -- in the workspace, one row per site from the window's first site on, and pass A sweeps those
  // chunks without rebuilding them.  The same rows the rebuild would produce: the same steps from the same vectors.
  // Built for the four-wave members in array mode (w2ResidentBuilt: what the host's planner asks): in the members of
  // six to eight waves the same lines -- two more copies of every wave's step function in the kernel -- moved the
  // register allocation of code that has nothing to do with them, and their SINGLE-chunk windows ran 3 - 14 % slower
  // (600 x 3000 list, K = 350 467 -> 482 ms, K = 500 612 -> 700 ms, K = 600 1150 -> 1238 ms); those members are compiled
  // from the lines they had.
  float4* const resBase = saveS + vecF4;
  constexpr bool kResidentBuilt = w2ResidentBuilt(NW, SEQ);
  const int nResident = kResidentBuilt ? p.residentChunks : 0;
  const unsigned laneOff = (unsigned)lane * (unsigned)sizeof(float4);
  const int C = p.chunk;
  const cfloat_p rowSets = (cfloat_p)p.rowSets;
  if (threadIdx.x < (unsigned)(KP / 4)) {
    piLds[threadIdx.x] = reinterpret_cast<const float4*>(p.pi)[threadIdx.x];
    if (MODE == kModePerPair) {
      coalLds[threadIdx.x] = reinterpret_cast<const float4*>(p.expCoal)[threadIdx.x];
    }
  }
  w2Barrier();
  const cfloat_p tCR = (cfloat_p)p.cR;
  const cfloat_p ghostMask = (cfloat_p)p.ghostMask;

  for (unsigned round = 0;; ++round) {
    unsigned g;
    if (MODE == kModeSums) {
      // one BATCH per workgroup and launch: workgroup i writes the sums of batch groupBase + i into plane i, and the
      // host adds the planes to the accumulator one after the other -- the reference's order, batch by batch
      // (HMM.cpp:1054-1073); the groups of a batch of more than 64 pairs are decoded in turn and continue the batch's
      // running sums (fsmc_kernels.h)
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
      if (threadIdx.x == 0) {
        groupLds = atomicAdd(&p.counters[p.groupBase], 1u);
      }
      w2Barrier();
      g = __builtin_amdgcn_readfirstlane(groupLds);
      w2Barrier(); // (groupLds is rewritten by the next round only after every wave has read it)
    }
    if (g >= (unsigned)p.nGroups) {
      break;
    }
    const cuint_p gw = (cuint_p)(p.groups + (size_t)g);
    const unsigned firstPair = gw[0];
    const int nPairsInGroup = (int)gw[1];
    const int from = (int)gw[2];
    const int to = (int)gw[3];
    const int scanFrom = (int)gw[4];
    const int aEnd = (MODE == kModeIbd) ? (int)gw[5] : to;
    // (recomputed where they are needed -- once per 64 sites, per record, per output row: no registers held for them)
    auto isValid = [&]() -> bool { return lane < nPairsInGroup; };
    auto pairIndex = [&]() -> unsigned { return firstPair + (lane < nPairsInGroup ? (unsigned)lane : 0u); };
    const int nA = aEnd - from;
    const int nChunks = (nA + C - 1) / C;
    const bool single = nChunks <= 1;
    // wave priority = how much of the group is left (fsmc_kernels.h, same place): of the two workgroups of a CU the one
    // with more in front of it is served first, so that they finish together (C4 3.60 -> 3.49 s, K = 192 9.27 -> 9.13 s)
    __builtin_amdgcn_s_setprio(3);
    // diagnostic builds only (-DFSMC_REGION_STAMPS): cycles per code region of this wave, flushed per group into
    // p.phaseCycles[8 + 30 * wave + region]: 0-9 / 10-19 the backward / forward step (see there), 20 / 21 what lies
    // between two backward steps / in front of a forward step (loop heads, row stores), 22 combine, 23 its sum,
    // 24 the consumer (25 its requests for the next rows, 26 the scan's sum, 27 decision, 28 its barrier)
    Diag cycW;
#if defined(FSMC_REGION_STAMPS)
    cycW.last = (unsigned)__builtin_readcyclecounter();
#endif

    int wordIdx = -1; // (wave-uniform: the 64-site block obsLds holds)
    auto obsClass = [&](const int q) -> int { // 0 het, 1 hom major, 2 hom minor (HMM.cpp:647-652)
      const int wi = q >> 6;
      if (__builtin_expect(wi != wordIdx, 0)) { // once per 64 sites
        const fsmc_pair pr = p.pairs[pairIndex()];
        const unsigned long long wa = p.haps[(size_t)pr.hap_a * p.W + wi];
        const unsigned long long wb = p.haps[(size_t)pr.hap_b * p.W + wi];
        obsLds[0][lane] = wa ^ wb;
        obsLds[1][lane] = wa & wb;
        wordIdx = wi;
        waitVm0();
      }
      const unsigned long long xw = obsLds[0][lane], aw = obsLds[1][lane];
      const int bit = q & 63;
      const int x = (int)((xw >> bit) & 1ull);
      const int t = (int)((aw >> bit) & 1ull);
      return x ? 0 : 1 + t;
    };
    // this wave's states of the three emission rows of site q into its ring slot (q & 1), by LDS-DMA
    auto stageEmis = [&](const int qIn) {
      // (wave-uniform by construction: the ring slot is an M0 value -- the sequence-mode call sites need it spelled out)
      const int q = SEQ ? __builtin_amdgcn_readfirstlane(qIn) : qIn;
#pragma unroll
      for (int i = 0; i < NLE; ++i) {
        const int idx = lane + i * kWave; // class * K4H + k4
        if (idx < E4H) {
          const int cls = idx / ERS, k4r = idx - cls * ERS;
          const int k4 = k4r < K4H ? k4r : K4H - 1; // (the padding lane)
          const float4* src = p.emis3 + (size_t)q * (NC * (KP / 4)) + (size_t)cls * (KP / 4) + h * K4H + k4;
          dmaToLds((gf32x4_p)src, &emisLds[h][q & 1][i * kWave]);
        }
      }
    };
    // the 64-site blocks the two slots of rowLds hold (wave-uniform); two slots, so that a block's indices stay readable
    // while a wave that is ahead loads the next block's
    int rowBlk0 = -1, rowBlk1 = -1;
    auto stepRowOf = [&](const int site) -> int {
      const int blk = site >> 6;
      const int slot = blk & 1;
      if (__builtin_expect(blk != (slot ? rowBlk1 : rowBlk0), 0)) { // once per 64 sites
        const int idx = blk * kWave + lane;
        rowLds[slot][lane] = p.stepRow[idx < p.S ? idx : p.S - 1];
        if (slot) {
          rowBlk1 = blk;
        } else {
          rowBlk0 = blk;
        }
        waitVm0();
        waitLgkm0();
        __builtin_amdgcn_wave_barrier();
      }
      return __builtin_amdgcn_readfirstlane(rowLds[slot][site & (kWave - 1)]);
    };
    auto rowSetOfRow = [&](const int row) -> cfloat_p { return rowSets + (size_t)row * (kRowSetParts * KP); };
    // emission rows requested one iteration ago have landed (the K4H row stores issued behind them may still be in
    // flight: vector memory operations retire in order)
    auto waitEmisRows = [&](const bool storesBehind) {
      constexpr unsigned n = (unsigned)K4H;
      if (storesBehind && n < 64) {
        __builtin_amdgcn_s_waitcnt(0x0F70 | (n & 15u) | ((n >> 4) << 14));
      } else {
        waitVm0();
      }
    };
    auto storeHalf = [&](float4* row, const float (&v)[KH]) FSMC_W2_INLINE { // row: wave-uniform address of the stored vector
      const gchar_p base = uniformPtr(row + halfF4);
#pragma unroll
      for (int k4 = 0; k4 < K4H; ++k4) {
        const f32x4 ov = {v[4 * k4], v[4 * k4 + 1], v[4 * k4 + 2], v[4 * k4 + 3]};
        __builtin_nontemporal_store(ov, rowSlot(base, k4, laneOff));
      }
    };
    auto loadHalf = [&](const float4* row, float (&v)[KH]) FSMC_W2_INLINE {
      const gchar_p base = uniformPtr(row + halfF4);
#pragma unroll
      for (int k4 = 0; k4 < K4H; ++k4) {
        const f32x4 ov = __builtin_nontemporal_load(rowSlot(base, k4, laneOff));
        v[4 * k4] = ov.x;
        v[4 * k4 + 1] = ov.y;
        v[4 * k4 + 2] = ov.z;
        v[4 * k4 + 3] = ov.w;
      }
    };
    auto fetchBeta = [&](const float4* row) {
      if constexpr (LAND) {
        const gchar_p base = uniformPtr(row + halfF4);
#pragma unroll
        for (int k4 = 0; k4 < K4H; ++k4) {
#if defined(__HIP_DEVICE_COMPILE__)
          __builtin_amdgcn_global_load_lds(rowSlot(base, k4, laneOff), &betaLds[h][k4 * kWave], 16, 0, 2 /* nt */);
#endif
        }
      }
    };
    // beta at the last site of the window: all ones, scaled (HMM.cpp:887-897): 1.0f / K for a state of the model
    // (K sequential additions of 1.0f are exact), +0 for a ghost
    auto betaInit = [&](float (&b)[KH]) FSMC_W2_INLINE {
      const float c = 1.0f / (float)K;
#pragma unroll
      for (int k = 0; k < KH; ++k) {
        b[k] = (h * KH + k < K) ? 1.0f * c : 0.f;
      }
    };
    // beta of site q -> beta of site q-1.  OUT: the row the step starts from -- beta of site q -- goes to `outRow` piece
    // by piece during the step (RowIO)
    auto betaStepInto = [&](float (&b)[KH], float (&w)[KH], const int q, auto moveRow, float4* outRow) FSMC_W2_INLINE {
      constexpr bool OUT = decltype(moveRow)::value;
      const int c = obsClass(q);
      const cfloat_p rsq = rowSetOfRow(SEQ ? __builtin_amdgcn_readfirstlane(p.rowSiteB[q]) : stepRowOf(q));
      const float4* eq = &emisLds[h][q & 1][c * ERS];
      const RowIO out = {OUT ? uniformPtr(outRow + halfF4) : (gchar_p) nullptr, laneOff, nullptr};
      FSMC_W2_ROLE(h, (beta_step_w2<NW, KH, H, true, OUT>(cx, b, w, rsq, eq, ghostMask, cycW, out)));
    };
    // a spare row of the workspace (checkpoint slot 0 is never a checkpoint): where an OUT step without a row of its
    // own to store writes
    float4* const spareRow = ckpt;
    // sequence mode: the un-normalised half-step across the gap (q-1, q), with the homozygous emission row of site q
    // (the fourth row of its ring slot)
    auto betaGapStep = [&](float (&b)[KH], float (&w)[KH], const int q) FSMC_W2_INLINE {
      const cfloat_p rsq = rowSetOfRow(__builtin_amdgcn_readfirstlane(p.rowGapB[q]));
      const float4* eq = &emisLds[h][q & 1][3 * ERS];
      FSMC_W2_ROLE(h, (beta_step_w2<NW, KH, H, false>(cx, b, w, rsq, eq, ghostMask, cycW, noRowIO())));
    };
    // the site step out of q = pos+1 (its rows are in the ring), then the half-step towards pos-1 unless pos is the
    // window start; the vector carried from site to site is the STORED one (after the half-step)
    auto betaSeqStep = [&](float (&b)[KH], float (&w)[KH], const int pos) FSMC_W2_INLINE {
      const int q = pos + 1;
      waitVm0(); // the rows of site q have landed
      __builtin_amdgcn_wave_barrier();
      if (pos > from) {
        stageEmis(pos); // into the slot of site pos + 2, whose steps are over
      }
      betaStepInto(b, w, q, std::false_type{}, nullptr);
      if (pos > from) {
        waitVm0();
        __builtin_amdgcn_wave_barrier();
        betaGapStep(b, w, pos);
      }
    };

    float w[KH];
    // ------------------------------------------------------------------ pass B
    {
      float b[KH];
      betaInit(b);
      int ckJ = (aEnd < to) ? nChunks : nChunks - 1;
      int ckPos = (aEnd < to) ? aEnd : from + ckJ * C;
      auto afterBeta = [&](const int pos) -> bool {
        if (single) {
          if (pos < aEnd) {
            storeHalf(chunkbuf + (size_t)(pos - from) * vecF4, b);
            return true;
          }
        } else if (__builtin_expect(pos == ckPos && ckJ >= 1, 0)) {
          storeHalf(ckpt + (size_t)ckJ * vecF4, b);
          ckJ -= 1;
          ckPos = from + ckJ * C;
          return true;
        }
        return false;
      };
      if constexpr (SEQ) {
        if (to - 1 > from) {
          stageEmis(to - 1);
          waitVm0();
          __builtin_amdgcn_wave_barrier();
          betaGapStep(b, w, to - 1);
        }
        afterBeta(to - 1);
        for (int pos = to - 2; pos >= from; --pos) {
          betaSeqStep(b, w, pos);
          afterBeta(pos);
        }
      } else {
        // Single-chunk windows keep every row of the alpha sweep's range: the row of site q rides out during the step
        // that starts from it (RowIO; a site beyond the sweep's range writes the spare row), the last one -- beta of the
        // window's first site -- behind the loop.  Chunked windows keep checkpoints only: a burst once per chunk.
        if (to - 2 >= from) {
          stageEmis(to - 1);
        }
        if (single) {
          bool stored = false; // (row stores issued behind the emission-row request of the iteration before)
          for (int pos = to - 2; pos >= from; --pos) {
            const int q = pos + 1;
            waitEmisRows(stored);
            __builtin_amdgcn_wave_barrier();
            if (pos - 1 >= from) {
              stageEmis(q - 1);
            }
            betaStepInto(b, w, q, std::true_type{}, q < aEnd ? chunkbuf + (size_t)(q - from) * vecF4 : spareRow);
            stored = true;
          }
          if (from < aEnd) {
            storeHalf(chunkbuf, b); // beta of the window's first site
          }
        } else if constexpr (kResidentBuilt) {
          // the sites whose rows stay: [from, resEnd); the lowest vector pass B has to reach: beta of the window's first
          // site if there are such rows, otherwise the first checkpoint (the first chunk is rebuilt from it)
          const int resEnd = (from + nResident * C < aEnd) ? from + nResident * C : aEnd;
          const int lowest = nResident > 0 ? from : from + C;
          bool stored = afterBeta(to - 1);
          if (to - 2 >= lowest) {
            stored = false; // (the request is behind the stores: wait for everything once)
          }
          int pos = to - 2;
          // above the resident rows: checkpoints only
          for (; pos >= lowest && pos + 1 >= resEnd; --pos) {
            const int q = pos + 1;
            waitEmisRows(stored);
            __builtin_amdgcn_wave_barrier();
            FSMC_END(cycW, 29);
            if (pos - 1 >= lowest) {
              stageEmis(q - 1);
            }
            betaStepInto(b, w, q, std::false_type{}, nullptr);
            stored = afterBeta(pos);
          }
          // the resident rows: the row of site q rides out during the step that starts from it (as in a single-chunk
          // window), the window's first behind the loop; no checkpoints down here -- these chunks are not rebuilt
          if (nResident > 0) {
            for (; pos >= from; --pos) {
              const int q = pos + 1;
              waitEmisRows(stored);
              __builtin_amdgcn_wave_barrier();
              if (pos - 1 >= from) {
                stageEmis(q - 1);
              }
              betaStepInto(b, w, q, std::true_type{}, resBase + (size_t)(q - from) * vecF4);
              stored = true;
            }
            storeHalf(resBase, b); // beta of the window's first site
          }
        } else {
          bool stored = afterBeta(to - 1);
          if (to - 2 >= from) {
            stored = false; // (the request is behind the stores: wait for everything once)
          }
          for (int pos = to - 2; pos >= from; --pos) {
            const int q = pos + 1;
            waitEmisRows(stored);
            __builtin_amdgcn_wave_barrier();
            FSMC_END(cycW, 29);
            if (pos - 1 >= from) {
              stageEmis(q - 1);
            }
            betaStepInto(b, w, q, std::false_type{}, nullptr);
            stored = afterBeta(pos);
          }
        }
      }
    }

    // ------------------------------------------------------------------ pass A
    int cur = 4;
    int segStart = 0;
    float acc = 0.f;
    float a[KH];
    auto emit = [&](const int s0, const int s1) { // wave 0 only (lane = pair)
      const unsigned idx = atomicAdd(&p.counters[1], 1u);
      float mean = 0.f, mapv = 0.f;
      if constexpr (TRACK) {
        segment_ages(K, p.ageThr, spsMem, tPi, tExpT, (p.flags & FSMC_WANT_MEAN) != 0, (p.flags & FSMC_WANT_MAP) != 0, mean,
                     mapv);
      }
      if (idx < p.recCap) {
        fsmc_ibd_record r;
        r.pair = pairIndex();
        r.start = s0;
        r.end = s1;
        r.prob = acc;
        r.post_mean = mean;
        r.map = mapv;
        p.recs[idx] = r;
      }
    };
    for (int j = 0; j < (nChunks > 0 ? nChunks : 0); ++j) {
      {
        const int left = (8 * (nChunks - j)) / (3 * nChunks);
        if (left >= 2) {
          __builtin_amdgcn_s_setprio(2);
        } else if (left == 1) {
          __builtin_amdgcn_s_setprio(1);
        } else {
          __builtin_amdgcn_s_setprio(0);
        }
      }
      const int lo = from + j * C;
      const int hi = (lo + C < aEnd) ? lo + C : aEnd;
      // the rows of this chunk: kept by pass B (a resident chunk), or rebuilt below into the chunk buffer
      const bool resident = kResidentBuilt && !single && j < nResident;
      float4* const cbuf = resident ? resBase + (size_t)(lo - from) * vecF4 : chunkbuf;
      if (!single && !resident) {
        if (j > 0) {
          storeHalf(saveA, a);
        }
        float b[KH];
        int pos;
        float4* pending = spareRow; // (array mode) where the row in b goes: it rides out during the next step
        if (hi == to) {
          betaInit(b);
          if constexpr (SEQ) {
            if (to - 1 > from) {
              stageEmis(to - 1);
              waitVm0();
              __builtin_amdgcn_wave_barrier();
              betaGapStep(b, w, to - 1);
            }
            storeHalf(chunkbuf + (size_t)(to - 1 - lo) * vecF4, b);
          } else {
            pending = chunkbuf + (size_t)(to - 1 - lo) * vecF4;
          }
          pos = to - 2;
        } else {
          loadHalf(ckpt + (size_t)(j + 1) * vecF4, b); // (the next chunk's row: not stored here)
          pos = hi - 1;
          if constexpr (SEQ) {
            stageEmis(hi); // the checkpoint is the stored vector of site hi: its rows next
          }
        }
        if constexpr (SEQ) {
          for (; pos >= lo; --pos) {
            betaSeqStep(b, w, pos);
            storeHalf(chunkbuf + (size_t)(pos - lo) * vecF4, b);
          }
        } else {
          if (pos >= lo) {
            stageEmis(pos + 1);
          }
          // (the row stores of the step before may stay in flight: only the emission rows requested in front of them
          //  must have landed -- a full wait here parked the wave for a store round trip at every site)
          // the row a step produces rides out during the step after it (RowIO); the chunk's first row behind the loop
          bool stored = false;
          for (; pos >= lo; --pos) {
            const int q = pos + 1;
            waitEmisRows(stored);
            __builtin_amdgcn_wave_barrier();
            FSMC_END(cycW, 29); // (diagnostic split of region 20: loop control + the wait for the emission rows)
            if (pos - 1 >= lo) {
              stageEmis(q - 1);
            }
            betaStepInto(b, w, q, std::true_type{}, pending);
            stored = true;
            pending = chunkbuf + (size_t)(pos - lo) * vecF4;
          }
          if (pending != spareRow) {
            storeHalf(pending, b);
          }
        }
        if (j > 0) {
          loadHalf(saveA, a);
        }
      }
      // this wave's own stores of the chunk's betas must have landed before its DMA reads them back
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      waitVm0();
      fetchBeta(cbuf);
      // sites whose emission rows the sweep of this chunk needs: sequence mode also takes the half-step out of the
      // chunk's last site, with the homozygous row of the site behind it
      const int stageEnd = SEQ ? (hi + 1 < to ? hi + 1 : to) : hi;
      stageEmis(lo);
      if (lo + 1 < stageEnd) {
        stageEmis(lo + 1);
      }
      waitVm0();
      for (int pos = lo; pos < hi; ++pos) {
        const int c = obsClass(pos);
        const float4* e = &emisLds[h][pos & 1][c * ERS];
        if (__builtin_expect(pos == from, 0)) {
          // alpha at the first site: pi * emission, scaled (HMM.cpp:736-747)
#pragma unroll
          for (int k4 = 0; k4 < K4H; ++k4) {
            const float4 ev = e[k4];
            const float4 pv = piLds[h * K4H + k4];
            w[4 * k4] = pv.x * ev.x;
            w[4 * k4 + 1] = pv.y * ev.y;
            w[4 * k4 + 2] = pv.z * ev.z;
            w[4 * k4 + 3] = pv.w * ev.w;
          }
          float total = 0.f;
          FSMC_W2_ROLE(h, (total = w2OrderedTotal<NW, KH, H>(cx, w, W2Rows<NW>::Step)));
          // (alpha_init multiplies by 1.0f / sum as well, HMM.cpp:744-747)
          w2Scale<KH>(a, w, total);
        } else {
          const cfloat_p rsp = rowSetOfRow(stepRowOf(pos));
          // this site's beta row is requested during the step (RowIO): the combine of the site before has released the
          // landing zone.  (The chunk's first row was requested in front of the loop: its step requests it once more,
          // the same bytes.  The sums consumer transposes its tile through the zone and requests the next row itself.)
          if constexpr (MODE != kModeSums && LAND) {
            const RowIO in = {uniformPtr(cbuf + (size_t)(pos - lo) * vecF4 + halfF4), laneOff, &betaLds[h][0]};
            FSMC_W2_ROLE(h, (alpha_step_w2<NW, KH, H, true, true>(cx, a, w, rsp, tCR, e, cycW, in)));
          } else {
            FSMC_W2_ROLE(h, (alpha_step_w2<NW, KH, H>(cx, a, w, rsp, tCR, e, cycW, noRowIO())));
          }
        }
        if constexpr (SEQ) {
          // what the reference's alpha buffer holds for this site: alpha after the un-normalised half-step across the gap
          // to the next site (HMM.cpp:764-767); the last site of the window keeps its alpha
          if (pos < to - 1) {
            waitVm0(); // the rows of site pos + 1 (requested a site ago) have landed
            __builtin_amdgcn_wave_barrier();
            const cfloat_p rsg = rowSetOfRow(__builtin_amdgcn_readfirstlane(p.rowGapF[pos + 1]));
            const float4* eg = &emisLds[h][(pos + 1) & 1][3 * ERS];
            FSMC_W2_ROLE(h, (alpha_step_w2<NW, KH, H, false>(cx, a, w, rsg, tCR, eg, cycW, noRowIO())));
          }
        }
        // combine with beta of this site (landed in LDS) and normalise (HMM.cpp:672-691)
        waitVm0();
        __builtin_amdgcn_wave_barrier();
        // (in blocks of sixteen states, the next block's reads in flight: left to itself the compiler issues all the
        //  landing-zone reads first -- a third K-vector of registers, and part of alpha went to scratch memory.  What
        //  orders a block's reads behind the products of the block before is an empty asm statement that takes those
        //  products as operands and clobbers memory; a scheduling barrier alone does not bind instruction selection)
        if constexpr (!LAND) {
          // no landing zone: this wave's part of the row out of the workspace (w is free between the step and the combine)
          loadHalf(cbuf + (size_t)(pos - lo) * vecF4, w);
#pragma unroll
          for (int k = 0; k < KH; k += 2) {
            const f32x2 av = {a[k], a[k + 1]}, bv = {w[k], w[k + 1]};
            const f32x2 q = pmul(av, bv);
            w[k] = q.x;
            w[k + 1] = q.y;
          }
        } else {
          constexpr int kCB4 = 4; // float4 per block
          float4 cb[kCB4], nb[kCB4];
#pragma unroll
          for (int j = 0; j < kCB4; ++j) {
            cb[j] = betaLds[h][j * kWave + lane];
          }
#pragma unroll
          for (int k4 = 0; k4 < K4H; k4 += kCB4) {
            if (k4 > 0) {
              const int k = 4 * (k4 - kCB4);
              FSMC_GCN_ASM("" ::"v"(w[k]), "v"(w[k + 1]), "v"(w[k + 2]), "v"(w[k + 3]), "v"(w[k + 4]), "v"(w[k + 5]),
                           "v"(w[k + 6]), "v"(w[k + 7]), "v"(w[k + 8]), "v"(w[k + 9]), "v"(w[k + 10]), "v"(w[k + 11]),
                           "v"(w[k + 12]), "v"(w[k + 13]), "v"(w[k + 14]), "v"(w[k + 15])
                           : "memory");
            }
            if (k4 + kCB4 < K4H) {
#pragma unroll
              for (int j = 0; j < kCB4; ++j) {
                nb[j] = betaLds[h][(k4 + kCB4 + j) * kWave + lane];
              }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < kCB4; ++j) {
              const float4 bv = cb[j];
              const int k = 4 * (k4 + j);
              const f32x2 a0 = {a[k], a[k + 1]}, a1 = {a[k + 2], a[k + 3]};
              const f32x2 b0 = {bv.x, bv.y}, b1 = {bv.z, bv.w};
              const f32x2 q0 = pmul(a0, b0), q1 = pmul(a1, b1);
              w[k] = q0.x;
              w[k + 1] = q0.y;
              w[k + 2] = q1.x;
              w[k + 3] = q1.y;
            }
#pragma unroll
            for (int j = 0; j < kCB4; ++j) {
              cb[j] = nb[j];
            }
          }
        }
        FSMC_END(cycW, 22);
        float sumq = 0.f;
        FSMC_W2_ROLE(h, (sumq = w2OrderedTotal<NW, KH, H>(cx, w, W2Rows<NW>::Comb)));
        const float cq = 1.0f / sumq;
        FSMC_END(cycW, 23);
        // every read of this site's ring slot has returned (the barriers above waited for lgkmcnt(0)): request the rows of
        // site pos + 2 (the next site's beta row rides in during its forward step)
        if (pos + 2 < stageEnd) {
          stageEmis(pos + 2);
        }
        FSMC_END(cycW, 25); // (the requests for the next beta row and emission rows)

        if (MODE == kModeSums) {
          // HMM::augmentSumOverPairs (HMM.cpp:1052-1081): per site and state, the batch's posteriors are summed over
          // pairs in batch order (local fp32 sum from 0.f).  Every wave transposes the tile of ITS states through its
          // landing zone (64 x 64 floats, row k rotated by k lanes: conflict-free both ways); lane j then owns state j.
          float* const tile = reinterpret_cast<float*>(&betaLds[h][0]);
          if (p.flags & FSMC_WANT_MAJOR_MINOR_SUMS) {
            clsLds[h][lane] = (unsigned char)c;
          }
          // (a member without landing zones: kTileStates states a turn through a tile of that many rows)
#pragma unroll
          for (int t0 = 0; t0 < KH; t0 += kTileStates) {
            if (t0 > 0) { // the walk of the turn before has read the tile
              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
              waitLgkm0();
              __builtin_amdgcn_wave_barrier();
            }
#pragma unroll
            for (int k = 0; k < kTileStates; ++k) {
              if (t0 + k < KH) {
                tile[k * kWave + ((lane + k) & (kWave - 1))] = w[t0 + k] * cq;
              }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            waitLgkm0();
            __builtin_amdgcn_wave_barrier();
            // lane j owns state j of every 64 of the turn's states (a wave of more than 64 states takes two rounds)
#pragma unroll
            for (int base = 0; base < kTileStates; base += kWave) {
              const int kTile = base + lane;    // row of the tile
              const int kLocal = t0 + kTile;     // state of this wave
              const int state = h * KH + kLocal;
              if (kTile < kTileStates && kLocal < KH && state < K) {
                float* acc = p.sums + (size_t)blockIdx.x * p.sumsSlot + (size_t)pos * K + state;
                float s = 0.f, s00 = 0.f, s01 = 0.f, s11 = 0.f;
                if (round > 0) { // a later group of the batch: the running sums of the pairs before (this wave wrote them)
                  if (p.flags & FSMC_WANT_SUMS) s = acc[0];
                  if (p.flags & FSMC_WANT_MAJOR_MINOR_SUMS) {
                    s00 = acc[p.sumsPlane];
                    s01 = acc[2 * p.sumsPlane];
                    s11 = acc[3 * p.sumsPlane];
                  }
                }
                // (sixteen pairs a turn: their tile values are read together, then added in batch order; the 00 / 01 /
                //  11 split adds +0.f to the sums a pair does not belong to -- fsmc_kernels.h, same place)
                auto walk = [&](auto splitTag) {
                  constexpr bool SPLIT = decltype(splitTag)::value;
                  constexpr int kWalk = 16;
                  auto add = [&](const float q, const int cv) {
                    s = s + q;
                    if constexpr (SPLIT) { // 0 het -> 01, 1 hom major -> 00, 2 hom minor -> 11
                      s11 = s11 + (cv == 2 ? q : 0.f);
                      s00 = s00 + (cv == 1 ? q : 0.f);
                      s01 = s01 + (cv == 0 ? q : 0.f);
                    }
                  };
                  int v = 0;
                  for (; v + kWalk <= nPairsInGroup; v += kWalk) {
                    float q[kWalk];
                    int cv[kWalk];
#pragma unroll
                    for (int i = 0; i < kWalk; ++i) {
                      q[i] = tile[kTile * kWave + ((v + i + kTile) & (kWave - 1))];
                      cv[i] = SPLIT ? (int)clsLds[h][v + i] : 0;
                    }
#pragma unroll
                    for (int i = 0; i < kWalk; ++i) {
                      add(q[i], cv[i]);
                    }
                  }
                  for (; v < nPairsInGroup; ++v) {
                    add(tile[kTile * kWave + ((v + kTile) & (kWave - 1))], SPLIT ? (int)clsLds[h][v] : 0);
                  }
                };
                if (p.flags & FSMC_WANT_MAJOR_MINOR_SUMS) {
                  walk(std::true_type{});
                } else {
                  walk(std::false_type{});
                }
                if (p.flags & FSMC_WANT_SUMS) acc[0] = s;
                if (p.flags & FSMC_WANT_MAJOR_MINOR_SUMS) {
                  acc[p.sumsPlane] = s00;
                  acc[2 * p.sumsPlane] = s01;
                  acc[3 * p.sumsPlane] = s11;
                }
              }
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          waitLgkm0();
          __builtin_amdgcn_wave_barrier();
          if (pos + 1 < hi) {
            fetchBeta(cbuf + (size_t)(pos + 1 - lo) * vecF4);
          }
        }

        if (MODE == kModePerPair) {
          // HMM::writePerPairOutput (HMM.cpp:1378-1409): mean = sum_k post*E[t_k] (k ascending from 0.f), MAP = first
          // strictly larger posterior -- both walk the states in order, hence the waves in order (three hand-overs)
          float mean = 0.f, best = 0.f;
          int arg = 0;
#pragma unroll
          for (int ph = 0; ph < NW; ++ph) {
            if (h == ph) {
              if (ph > 0) {
                mean = cx.mail[(W2Rows<NW>::Mean + ph - 1) * kWave + lane];
                best = cx.mail[(W2Rows<NW>::Step + ph - 1) * kWave + lane];
                arg = __float_as_int(cx.mail[(W2Rows<NW>::Comb + ph - 1) * kWave + lane]);
              }
#pragma unroll
              for (int k4 = 0; k4 < K4H; ++k4) {
                const float4 tc = coalLds[h * K4H + k4];
                const float t4[4] = {tc.x, tc.y, tc.z, tc.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                  const float post = w[4 * k4 + i] * cq;
                  mean = mean + post * t4[i];
                  if (best < post) {
                    arg = h * KH + 4 * k4 + i;
                    best = post;
                  }
                }
              }
              if (ph < NW - 1) {
                cx.mail[(W2Rows<NW>::Mean + ph) * kWave + lane] = mean;
                cx.mail[(W2Rows<NW>::Step + ph) * kWave + lane] = best;
                cx.mail[(W2Rows<NW>::Comb + ph) * kWave + lane] = __int_as_float(arg);
              }
            }
            if (ph < NW - 1) {
              w2Barrier();
            }
          }
          if (h == NW - 1 && isValid()) {
            if (p.ppMean) p.ppMean[(size_t)pairIndex() * p.S + pos] = mean;
            if (p.ppMap) p.ppMap[(size_t)pairIndex() * p.S + pos] = arg;
          }
        }

        if (MODE == kModeDump) {
          float* out = p.dumpOut + p.dumpOffsets[g] + (size_t)(pos - from) * K * kWave + lane;
#pragma unroll
          for (int k = 0; k < KH; ++k) {
            if (h * KH + k < K) {
              out[(size_t)(h * KH + k) * kWave] = isValid() ? w[k] * cq : 0.f;
            }
          }
        }

        if (MODE == kModeIbd) {
          if (pos >= scanFrom) {
            // sum over the states below the threshold, k ascending from 0.f (HMM.cpp:1207-1224): wave 0 first, the next
            // waves join only when the threshold reaches their states (uniform over the launch)
            const unsigned nPost = p.stateThr;
            int nScanWaves = 1;
#pragma unroll
            for (int i = 1; i < NW; ++i) {
              nScanWaves += nPost > (unsigned)(i * KH) ? 1 : 0;
            }
            float s = 0.f;
            // (this wave's states below the threshold: scanBlocks, fsmc_kernels.h)
            auto partial = [&](float s0) -> float {
              const unsigned nLocal = nPost - (unsigned)(h * KH) < (unsigned)KH ? nPost - (unsigned)(h * KH) : (unsigned)KH;
              scanBlocks<KH, KH, K4H>(w, s0, cq, launderScalar((unsigned)__builtin_amdgcn_readfirstlane((int)nLocal)));
              return s0;
            };
#pragma unroll
            for (int ph = 0; ph < NW; ++ph) {
              if (ph < nScanWaves) {
                if (h == ph) {
                  s = partial(ph == 0 ? 0.f : cx.mail[(W2Rows<NW>::Scan + ph - 1) * kWave + lane]);
                  if (nScanWaves > 1) {
                    cx.mail[(W2Rows<NW>::Scan + ph) * kWave + lane] = s;
                  }
                }
                if (nScanWaves > 1) {
                  w2Barrier();
                }
              }
            }
            if (nScanWaves > 1) {
              s = cx.mail[(W2Rows<NW>::Scan + nScanWaves - 1) * kWave + lane];
            }
            FSMC_END(cycW, 26); // (the scan's sum)
            // the scan's state machine runs in wave 0 (lane = pair)
            int level = 4;
            bool opening = false;
            bool closing = false; // a change of level (or a drop below every threshold) closes the open segment at pos-1
            if (h == 0) {
              level = s >= p.thr[0] ? 0 : s >= p.thr[1] ? 1 : s >= p.thr[2] ? 2 : s >= p.thr[3] ? 3 : 4;
              opening = level != 4 && level != cur;
              closing = isValid() && cur != 4 && level != cur;
            }
            // the other waves hold states the segment ages read: they need the decision, and wave 0 their sums
            const bool upperAges = TRACK && p.ageThr > (unsigned)KH;
            if (upperAges) {
              const int row = W2Rows<NW>::Level + (pos & 1); // (two rows in turn: one barrier a site is enough)
              if (h == 0) {
                cx.mail[row * kWave + lane] = __int_as_float(level | (opening ? 8 : 0) | (closing ? 16 : 0));
              } else {
                // this wave's sums of the sites before are in memory before wave 0 may read them -- but not the request
                // for the emission values of site pos + 2 issued a moment ago, behind those stores (vector memory
                // operations retire in order: "at most that many outstanding" means the stores are done)
                constexpr unsigned nE = (unsigned)NLE;
                static_assert(nE < 16, "one-digit vmcnt");
                if (pos + 2 < stageEnd) {
                  __builtin_amdgcn_s_waitcnt(0x0F70 | nE);
                } else {
                  waitVm0();
                }
              }
              FSMC_END(cycW, 27); // (decision, or the wait for this wave's sums)
              w2Barrier();
              FSMC_END(cycW, 28); // (the barrier that hands the decision over)
              if (h != 0) {
                const int v = __float_as_int(cx.mail[row * kWave + lane]);
                level = v & 7;
                opening = (v & 8) != 0;
                closing = (v & 16) != 0;
              }
            }
            if (h == 0 && __builtin_expect(closing, 0)) {
              emit(segStart, pos - 1);
            }
            if (upperAges && __builtin_expect(__ballot(closing) != 0ull, 0)) {
              // wave 0 reads the other waves' sums while it closes a segment: they wait before they add this site
              // (the same lanes close in every wave's copy of the decision: the branch is uniform over the workgroup)
              w2Barrier();
            }
            if constexpr (TRACK) {
              // per-state posterior sums of the open segment (HMM.cpp:1212-1229), each wave its own states
              if (level != 4 && (h == 0 || upperAges)) {
                const gchar_p spsBase = uniformPtr(saveS + halfF4);
                // (thresholds the compiler cannot prove loop-invariant: fsmc_kernels.h, same place)
                const unsigned nAgeL = launderScalar(p.ageThr), nPostL = launderScalar(nPost);
                // A lane that opens a segment at this site clears its column first (once per segment: 0.f + x is x, and
                // a wave's own store to an address is what its next load from it returns), so the accumulation has no
                // select; four blocks (sixteen states) per round trip to L2, their loads out together; products and sums
                // two states an instruction -- fsmc_kernels.h, same place.  (One block a round trip, as this loop was
                // written first, parked the wave sixteen times per site.)
                if (__builtin_expect(opening, 0)) {
#pragma unroll
                  for (int k4 = 0; k4 < K4H; ++k4) {
                    if ((unsigned)(h * KH + 4 * k4) >= nAgeL) {
                      break;
                    }
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    *rowSlot(spsBase, k4, laneOff) = z;
                  }
                }
                constexpr int kG = 4;
                static_assert(K4H % kG == 0, "whole rounds");
#pragma unroll
                for (int g4 = 0; g4 < K4H; g4 += kG) {
                  if ((unsigned)(h * KH + 4 * g4) >= nAgeL) {
                    break;
                  }
                  f32x4 sv[kG];
#pragma unroll
                  for (int j = 0; j < kG; ++j) {
                    sv[j] = *rowSlot(spsBase, g4 + j, laneOff);
                  }
#pragma unroll
                  for (int j = 0; j < kG; ++j) {
                    const int k4 = g4 + j;
                    const float sc = ((unsigned)(h * KH + 4 * k4) < nPostL) ? 1.0f : cq;
                    const f32x2 scv = {sc, sc};
                    const f32x2 w01 = {w[4 * k4], w[4 * k4 + 1]}, w23 = {w[4 * k4 + 2], w[4 * k4 + 3]};
                    const f32x2 s01 = {sv[j].x, sv[j].y}, s23 = {sv[j].z, sv[j].w};
                    const f32x2 r01 = padd(s01, pmul(w01, scv)), r23 = padd(s23, pmul(w23, scv));
                    const f32x4 o = {r01.x, r01.y, r23.x, r23.y};
                    *rowSlot(spsBase, k4, laneOff) = o;
                  }
                }
              }
            }
            if (h == 0) {
              acc = (level == 4) ? 0.f : (opening ? s : acc + s);
              if (opening) {
                segStart = pos;
              }
              cur = level;
            }
            if (__builtin_expect(pos == aEnd - 1, 0)) {
              // the last site of the scan window closes the open segment (the other waves' sums of this very site first)
              if (upperAges) {
                if (h != 0) {
                  waitVm0();
                }
                w2Barrier();
              }
              if (h == 0 && isValid() && cur != 4) {
                emit(segStart, pos);
              }
            }
          }
        }
        FSMC_END(cycW, 24);
      }
    }
#if defined(FSMC_REGION_STAMPS)
    if (lane == 0 && p.phaseCycles) {
#pragma unroll
      for (int r = 0; r < kDiagRegions; ++r) {
        atomicAdd(&p.phaseCycles[8 + kDiagRegions * h + r], (unsigned long long)cycW.acc[r]);
      }
    }
#endif
    // the next group reuses the workspace slot and the mailbox: everything of this one is over in both waves
    waitVm0();
    w2Barrier();
  }
}

} // namespace fsmc
