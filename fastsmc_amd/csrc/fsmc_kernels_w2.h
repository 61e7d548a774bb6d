// fsmc_kernels_w2.h -- the decode kernel for wide models (128 < K <= 256; K = 256 is BASELINE.json config 4) with
// lane = pair and SEVERAL WAVES per group.
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

constexpr int kW2NW = 4;    // waves per group
// mailbox rows (64 floats each): carries of the recurrences per boundary, partial sums per wave
constexpr int kW2RowT = 0, kW2RowBU = 3, kW2RowBL = 6, kW2RowC = 0, kW2RowAU = 3; // (+ boundary 0..2)
#if defined(FSMC_W2_DIAG_OLD_LDS) // layout experiment
constexpr int kW2RowStep = 9, kW2RowComb = 13, kW2RowLevel = 21;
constexpr int kW2RowScan = 17;
constexpr int kW2RowMean = kW2RowT;
constexpr int kW2Mail = 23;
#else
constexpr int kW2RowStep = 9, kW2RowComb = 13, kW2RowLevel = 17;
constexpr int kW2RowScan = kW2RowStep; // (the scan's partial sums follow the combine: the step's rows are free then)
constexpr int kW2RowMean = kW2RowT;    // (kModePerPair: likewise the carries' rows)
constexpr int kW2Mail = 19;
#endif
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

struct W2Ctx {
  float* mail;  // [kW2Mail][64] in LDS, shared by the waves of the group
  int lane;
  int h;        // which wave: states [KH*h, KH*h + KH)
  bool hi;      // a wave of the upper half: descending pass first
};

// Sum of all states in state order (HmmUtils.cpp:121-128): wave 0 adds its states from 0.f, every next wave continues.
// NW barriers; returns the total in every wave.  `row`: first of NW mailbox rows.
template <int KH, int H>
__device__ __forceinline__ float w2OrderedTotal(const W2Ctx& cx, const float (&v)[KH], const int row)
{
#pragma unroll
  for (int ph = 0; ph < kW2NW; ++ph) {
    if (H == ph) {
      float s = ph == 0 ? 0.f : cx.mail[(row + ph - 1) * kWave + cx.lane];
#pragma unroll
      for (int k = 0; k < KH; ++k) {
        s = s + v[k];
      }
      cx.mail[(row + ph) * kWave + cx.lane] = s;
    }
    if (ph == kW2NW - 1) {
      w2Barrier();
    } else {
      w2SumBarrier();
    }
  }
  return cx.mail[(row + kW2NW - 1) * kWave + cx.lane];
}

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
template <int KH, int H, bool SCALE = true>
__device__ __forceinline__ void beta_step_w2(const W2Ctx& cx, float (&b)[KH], float (&w)[KH], cfloat_p rs,
                                             const float4* e, cfloat_p ghostMask)
{
  constexpr int KP = kW2NW * KH;
  constexpr int kLines = KH / 16; // 64-byte lines of this wave's part of a table row
  static_assert(KH % kWBWide == 0 && KH % 16 == 0 && kLines <= 4, "whole operand blocks and lines");
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
#pragma unroll
      for (int i = BS - 1; i >= 0; --i) {
        const int k = blk * BS + i;
        const float tNext = (i == BS - 1) ? tAbove : T[i + 1];
        const float bu = tNext + rr[i] * buAbove;
        buAbove = bu;
        w[k] = accumulate ? w[k] + bu : bu;
      }
      tAbove = T[0];
    }
    if (H > 0) { // carry for the states below: (T, BU) of this wave's first state
      cx.mail[(kW2RowT + H - 1) * kWave + cx.lane] = tAbove;
      cx.mail[(kW2RowBU + H - 1) * kWave + cx.lane] = buAbove;
    }
  };
  // ---- ascending pass: BL[k] = BL[k-1] + B[k-1]*vec[k-1];  x[k] = BL[k] + D[k]*vec[k]
  // wave 0 runs it in phase 0 (BL[0] = 0; it also forms vec), wave 1 in phase 1 from wave 0's BL of state KH
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
    } else {
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
        } else {
          landed(nmk);
          mk = nmk;
        }
      } else {
        landed(d, bt);
        heldRow<kLines - 1>(td);
        heldRow<kLines - 1>(tbt);
        if (!first) {
          landed(mk);
        }
      }
      if (blk + 1 < NB) {
        nd = LD<BS, false>::loadAt(rsw, kRowD * KP + (blk + 1) * BS);
        nbt = LD<BS, false>::loadAt(rsw, kRowB * KP + (blk + 1) * BS);
        if (first) {
          nem = readEmis<BS>(e, blk + 1);
        } else {
          nmk = LD<BS, false>::loadAt(gmw, (blk + 1) * BS);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
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
          x = pmul(x, pairOf(mk, i));
        }
        w[k] = x.x;
        w[k + 1] = x.y;
        BL = bl.y + bv.y;
      }
    }
    if (H < kW2NW - 1) {
      cx.mail[(kW2RowBL + H) * kWave + cx.lane] = BL; // BL of the next wave's first state
    }
  };
#pragma unroll
  for (int ph = 0; ph < kW2NW; ++ph) {
    if (H == ph) { // the ascending pass reaches this wave
      const float blIn = ph == 0 ? 0.f : cx.mail[(kW2RowBL + ph - 1) * kWave + cx.lane];
      if ((H >= kW2NW / 2)) {
        ascending(std::integral_constant<int, kWB>{}, blIn, false); // three operand rows: the smaller blocks
      } else {
        ascending(std::integral_constant<int, kWBWide>{}, blIn, true);
      }
    }
    if (H == kW2NW - 1 - ph) { // the descending pass reaches this wave
      const float tIn = ph == 0 ? 0.f : cx.mail[(kW2RowT + H) * kWave + cx.lane];
      const float buIn = ph == 0 ? 0.f : cx.mail[(kW2RowBU + H) * kWave + cx.lane];
      if ((H >= kW2NW / 2)) {
        descending(std::integral_constant<int, kWBWide>{}, tIn, buIn, false);
      } else {
        descending(std::integral_constant<int, kWBWide>{}, tIn, buIn, true);
      }
    }
    w2PhaseBarrier();
  }
  if constexpr (SCALE) {
    const float total = w2OrderedTotal<KH, H>(cx, w, kW2RowStep);
    w2Scale<KH>(b, w, total);
  } else { // the un-normalised half-step of sequence mode (HMM.cpp:915-922)
#pragma unroll
    for (int k = 0; k < KH; ++k) {
      b[k] = w[k];
    }
  }
}

// One forward step (HMM.cpp:799-830) + scaling.  a: this wave's half of alpha of site pos-1 on entry, of pos on exit.
template <int KH, int H, bool SCALE = true>
__device__ __forceinline__ void alpha_step_w2(const W2Ctx& cx, float (&a)[KH], float (&w)[KH], cfloat_p rs, cfloat_p cR,
                                              const float4* e)
{
  constexpr int KP = kW2NW * KH;
  constexpr int NBF = KH / kWBF;
  constexpr int kLines = KH / 16;
  static_assert(KH % kWBF == 0 && KH % 16 == 0 && kLines <= 4, "whole operand blocks and lines");
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
    SVec d, u, c4, bt, nd, nu, nc, nbt;
    EmisBlk<kWBF> em, nem;
    d = LD<kWBF, false>::loadAt(rsw, kRowD * KP);
    u = LD<kWBF, false>::loadAt(rsw, kRowU * KP);
    c4 = LD<kWBF, false>::loadAt(crw, 0);
    Touched td, tu, tb;
    touchRow<1, kLines - 1>(td, rsw, kRowD * KP);
    touchRow<1, kLines - 1>(tu, rsw, kRowU * KP);
    if (complete) {
      bt = LD<kWBF, false>::loadAt(rsw, kRowB * KP);
      touchRow<1, kLines - 1>(tb, rsw, kRowB * KP);
      em = readEmis<kWBF>(e, 0);
    }
    float AU = auIn;
#pragma unroll
    for (int blk = 0; blk < NBF; ++blk) {
      FSMC_WAIT_OPERANDS(dummy);
      if (blk > 0) {
        if (complete) {
          landed(nd, nu, nc, nbt);
          bt = nbt;
          em = nem;
        } else {
          landed(nd, nu);
          landed(nc);
        }
        d = nd;
        u = nu;
        c4 = nc;
      } else {
        if (complete) {
          landed(d, u, c4, bt);
          heldRow<kLines - 1>(tb);
        } else {
          landed(d, u);
          landed(c4);
        }
        heldRow<kLines - 1>(td);
        heldRow<kLines - 1>(tu);
      }
      if (blk + 1 < NBF) {
        nd = LD<kWBF, false>::loadAt(rsw, kRowD * KP + (blk + 1) * kWBF);
        nu = LD<kWBF, false>::loadAt(rsw, kRowU * KP + (blk + 1) * kWBF);
        nc = LD<kWBF, false>::loadAt(crw, (blk + 1) * kWBF);
        if (complete) {
          nbt = LD<kWBF, false>::loadAt(rsw, kRowB * KP + (blk + 1) * kWBF);
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
          const f32x2 ac = {w[k], w[k + 1]};
          const f32x2 bw = pmul(pairOf(bt, i), ac);
          term = padd(term, bw);
          term = pmul(em.pair(i), term);
        }
        w[k] = term.x;
        w[k + 1] = term.y;
        AU = ua.y + c4[i + 1] * au.y; // AU of state k+2
      }
    }
    if (H < kW2NW - 1) {
      cx.mail[(kW2RowAU + H) * kWave + cx.lane] = AU; // AU of the next wave's first state
    }
  };
  // ---- suffix sums, descending: alphaC[k] = alphaC[k+1] + alpha[k] (HMM.cpp:799-814)
  // upper half: operand-free, w[k] = alphaC of the state above k (what the B term of state k needs); the model's last
  // state has no state above it: its slot is 0 and B*0 leaves its term unchanged (HMM.cpp:823-826)
  auto suffix = [&](const float cIn) {
    float c = cIn;
#pragma unroll
    for (int k = KH - 1; k >= 0; --k) {
      w[k] = c;
      c = c + a[k];
    }
    cx.mail[(kW2RowC + H - 1) * kWave + cx.lane] = c; // alphaC of this wave's first state (H >= 2 here)
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
      cx.mail[(kW2RowC + H - 1) * kWave + cx.lane] = c;
    }
  };
#pragma unroll
  for (int ph = 0; ph < kW2NW; ++ph) {
    if (H == ph) { // the AU recurrence reaches this wave
      const float auIn = ph == 0 ? 0.f : cx.mail[(kW2RowAU + ph - 1) * kWave + cx.lane];
      if ((H >= kW2NW / 2)) {
        ascending(auIn, true);
      } else {
        ascending(auIn, false);
      }
    }
    if (H == kW2NW - 1 - ph) { // the suffix sums reach this wave
      const float cIn = ph == 0 ? 0.f : cx.mail[(kW2RowC + H) * kWave + cx.lane];
      if ((H >= kW2NW / 2)) {
        suffix(cIn);
      } else {
        finish(cIn);
      }
    }
    w2PhaseBarrier();
  }
  if constexpr (SCALE) {
    const float total = w2OrderedTotal<KH, H>(cx, w, kW2RowStep);
    w2Scale<KH>(a, w, total);
  } else { // the un-normalised half-step of sequence mode (HMM.cpp:760-767)
#pragma unroll
    for (int k = 0; k < KH; ++k) {
      a[k] = w[k];
    }
  }
}

// The wave's role is a compile-time parameter of the step functions (its phases are then straight-line code); the
// kernel branches on the wave number once per call.
#define FSMC_W2_ROLE(h, CALL)                                                                                          \
  do {                                                                                                                 \
    switch (h) {                                                                                                       \
    case 0: { constexpr int H = 0; CALL; } break;                                                                      \
    case 1: { constexpr int H = 1; CALL; } break;                                                                      \
    case 2: { constexpr int H = 2; CALL; } break;                                                                      \
    default: { constexpr int H = 3; CALL; } break;                                                                     \
    }                                                                                                                  \
  } while (0)

// Work item = one group of <= 64 pairs; workgroup = kW2NW waves; two workgroups per CU (the landing zones fill LDS).
// SEQ: sequence mode (two steps per site, a fourth emission row per site: fsmc_kernels.h) -- the same schedule as there.
template <int KH, int MODE, bool TRACK, bool SEQ = false>
__global__ __launch_bounds__(kW2NW * kWave, 2) void decode_kernel_w2(const KParams p)
{
  static_assert(MODE == kModeIbd || MODE == kModeDump || MODE == kModeSums || MODE == kModePerPair,
                "the consumers of the wave-group kernel");
  constexpr int KP = kW2NW * KH;
  constexpr int K4H = KH / 4;        // float4 per lane of this wave's part of a K-vector
  constexpr int NC = SEQ ? 4 : 3;    // emission rows per site: three observation classes (+ the gap's homozygous row)
  constexpr int E4H = NC * K4H;      // float4 of one site's emission values of this wave's states
  constexpr int NLE = (E4H + kWave - 1) / kWave;
  __shared__ float4 emisLds[kW2NW][2][E4H];       // [wave][ring slot][class * K4H + k4]
  __shared__ float4 betaLds[kW2NW][K4H * kWave];  // [wave]: landing zone of the next site's beta row (its part)
  __shared__ float mailLds[kW2Mail * kWave];
  __shared__ float4 piLds[KP / 4];   // initialStateProb, zero padded
  __shared__ float4 coalLds[MODE == kModePerPair ? KP / 4 : 1]; // kModePerPair: expected coalescence times, zero padded
  __shared__ unsigned groupLds;
  __shared__ unsigned char clsLds[kW2NW][kWave]; // kModeSums: observation class of every pair at the current site

  const int lane = threadIdx.x & (kWave - 1);
  // (wave-uniform BY CONSTRUCTION: as a scalar the compiler branches on it instead of predicating both roles)
  const int h = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const W2Ctx cx = {mailLds, lane, h, h >= kW2NW / 2};
  const int K = p.K; // states of the model, <= KP
  const cfloat_p tPi = (cfloat_p)p.pi, tExpT = (cfloat_p)p.expT;
  const size_t vecF4 = (size_t)(KP / 4) * kWave; // float4 per stored K-vector of the group
  const size_t halfF4 = (size_t)h * K4H * kWave;  // this wave's half inside a stored vector
  float4* const chunkbuf = p.ws + (size_t)blockIdx.x * p.wsSlot;
  float4* const ckpt = chunkbuf + (size_t)p.chunkRows * vecF4;
  float4* const saveA = ckpt + (size_t)(p.maxChunks + 2) * vecF4;
  float4* const saveS = saveA + vecF4;
  float4* const spsMem = saveS + lane; // this lane's column of the per-state posterior sums (all states)
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
      // one group per workgroup and launch: workgroup i writes the batch sums of group groupBase + i into plane i, and
      // the host adds the planes to the accumulator one after the other -- the reference's order, batch by batch
      // (HMM.cpp:1054-1073)
      g = round == 0 ? (unsigned)p.groupBase + blockIdx.x : (unsigned)p.nGroups;
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
    const bool valid = lane < nPairsInGroup;
    const unsigned pairIdx = firstPair + (valid ? (unsigned)lane : 0u);
    const fsmc_pair pr = p.pairs[pairIdx];
    const unsigned long long* rowA = p.haps + (size_t)pr.hap_a * p.W;
    const unsigned long long* rowB = p.haps + (size_t)pr.hap_b * p.W;
    const int nA = aEnd - from;
    const int nChunks = (nA + C - 1) / C;
    const bool single = nChunks <= 1;

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
    // this wave's states of the three emission rows of site q into its ring slot (q & 1), by LDS-DMA
    auto stageEmis = [&](const int qIn) {
      // (wave-uniform by construction: the ring slot is an M0 value -- the sequence-mode call sites need it spelled out)
      const int q = SEQ ? __builtin_amdgcn_readfirstlane(qIn) : qIn;
#pragma unroll
      for (int i = 0; i < NLE; ++i) {
        const int idx = lane + i * kWave; // class * K4H + k4
        if (idx < E4H) {
          const int cls = idx / K4H, k4 = idx - cls * K4H;
          const float4* src = p.emis3 + (size_t)q * (NC * (KP / 4)) + (size_t)cls * (KP / 4) + h * K4H + k4;
          dmaToLds((gf32x4_p)src, &emisLds[h][q & 1][i * kWave]);
        }
      }
    };
    int rowBlk = -1;
    int rowVec = 0;
    auto stepRowOf = [&](const int site) -> int {
      const int blk = site >> 6;
      if (__builtin_expect(blk != rowBlk, 0)) {
        const int idx = blk * kWave + lane;
        rowVec = p.stepRow[idx < p.S ? idx : p.S - 1];
        rowBlk = blk;
        waitVm0();
      }
      return __builtin_amdgcn_readlane(rowVec, site & (kWave - 1));
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
    auto storeHalf = [&](float4* row, const float (&v)[KH]) { // row: wave-uniform address of the stored vector
      const gchar_p base = uniformPtr(row + halfF4);
#pragma unroll
      for (int k4 = 0; k4 < K4H; ++k4) {
        const f32x4 ov = {v[4 * k4], v[4 * k4 + 1], v[4 * k4 + 2], v[4 * k4 + 3]};
        __builtin_nontemporal_store(ov, rowSlot(base, k4, laneOff));
      }
    };
    auto loadHalf = [&](const float4* row, float (&v)[KH]) {
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
      const gchar_p base = uniformPtr(row + halfF4);
#pragma unroll
      for (int k4 = 0; k4 < K4H; ++k4) {
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_global_load_lds(rowSlot(base, k4, laneOff), &betaLds[h][k4 * kWave], 16, 0, 2 /* nt */);
#endif
      }
    };
    // beta at the last site of the window: all ones, scaled (HMM.cpp:887-897): 1.0f / K for a state of the model
    // (K sequential additions of 1.0f are exact), +0 for a ghost
    auto betaInit = [&](float (&b)[KH]) {
      const float c = 1.0f / (float)K;
#pragma unroll
      for (int k = 0; k < KH; ++k) {
        b[k] = (h * KH + k < K) ? 1.0f * c : 0.f;
      }
    };
    auto betaStepInto = [&](float (&b)[KH], float (&w)[KH], const int q) { // beta of site q -> beta of site q-1
      const int c = obsClass(q);
      const cfloat_p rsq = rowSetOfRow(SEQ ? __builtin_amdgcn_readfirstlane(p.rowSiteB[q]) : stepRowOf(q));
      const float4* eq = &emisLds[h][q & 1][c * K4H];
      FSMC_W2_ROLE(h, (beta_step_w2<KH, H>(cx, b, w, rsq, eq, ghostMask)));
    };
    // sequence mode: the un-normalised half-step across the gap (q-1, q), with the homozygous emission row of site q
    // (the fourth row of its ring slot)
    auto betaGapStep = [&](float (&b)[KH], float (&w)[KH], const int q) {
      const cfloat_p rsq = rowSetOfRow(__builtin_amdgcn_readfirstlane(p.rowGapB[q]));
      const float4* eq = &emisLds[h][q & 1][3 * K4H];
      FSMC_W2_ROLE(h, (beta_step_w2<KH, H, false>(cx, b, w, rsq, eq, ghostMask)));
    };
    // the site step out of q = pos+1 (its rows are in the ring), then the half-step towards pos-1 unless pos is the
    // window start; the vector carried from site to site is the STORED one (after the half-step)
    auto betaSeqStep = [&](float (&b)[KH], float (&w)[KH], const int pos) {
      const int q = pos + 1;
      waitVm0(); // the rows of site q have landed
      __builtin_amdgcn_wave_barrier();
      if (pos > from) {
        stageEmis(pos); // into the slot of site pos + 2, whose steps are over
      }
      betaStepInto(b, w, q);
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
        bool stored = afterBeta(to - 1);
        if (to - 2 >= from) {
          stageEmis(to - 1);
          stored = false; // (the request is behind the stores: wait for everything once)
        }
        for (int pos = to - 2; pos >= from; --pos) {
          const int q = pos + 1;
          waitEmisRows(stored);
          __builtin_amdgcn_wave_barrier();
          if (pos - 1 >= from) {
            stageEmis(q - 1);
          }
          betaStepInto(b, w, q);
          stored = afterBeta(pos);
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
        if (j > 0) {
          storeHalf(saveA, a);
        }
        float b[KH];
        int pos;
        if (hi == to) {
          betaInit(b);
          if constexpr (SEQ) {
            if (to - 1 > from) {
              stageEmis(to - 1);
              waitVm0();
              __builtin_amdgcn_wave_barrier();
              betaGapStep(b, w, to - 1);
            }
          }
          storeHalf(chunkbuf + (size_t)(to - 1 - lo) * vecF4, b);
          pos = to - 2;
        } else {
          loadHalf(ckpt + (size_t)(j + 1) * vecF4, b);
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
          for (; pos >= lo; --pos) {
            const int q = pos + 1;
            waitVm0();
            __builtin_amdgcn_wave_barrier();
            if (pos - 1 >= lo) {
              stageEmis(q - 1);
            }
            betaStepInto(b, w, q);
            storeHalf(chunkbuf + (size_t)(pos - lo) * vecF4, b);
          }
        }
        if (j > 0) {
          loadHalf(saveA, a);
        }
      }
      // this wave's own stores of the chunk's betas must have landed before its DMA reads them back
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      waitVm0();
      fetchBeta(chunkbuf);
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
        const float4* e = &emisLds[h][pos & 1][c * K4H];
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
          FSMC_W2_ROLE(h, (total = w2OrderedTotal<KH, H>(cx, w, kW2RowStep)));
          // (alpha_init multiplies by 1.0f / sum as well, HMM.cpp:744-747)
          w2Scale<KH>(a, w, total);
        } else {
          const cfloat_p rsp = rowSetOfRow(stepRowOf(pos));
          FSMC_W2_ROLE(h, (alpha_step_w2<KH, H>(cx, a, w, rsp, tCR, e)));
        }
        if constexpr (SEQ) {
          // what the reference's alpha buffer holds for this site: alpha after the un-normalised half-step across the gap
          // to the next site (HMM.cpp:764-767); the last site of the window keeps its alpha
          if (pos < to - 1) {
            waitVm0(); // the rows of site pos + 1 (requested a site ago) have landed
            __builtin_amdgcn_wave_barrier();
            const cfloat_p rsg = rowSetOfRow(__builtin_amdgcn_readfirstlane(p.rowGapF[pos + 1]));
            const float4* eg = &emisLds[h][(pos + 1) & 1][3 * K4H];
            FSMC_W2_ROLE(h, (alpha_step_w2<KH, H, false>(cx, a, w, rsg, tCR, eg)));
          }
        }
        // combine with beta of this site (landed in LDS) and normalise (HMM.cpp:672-691)
        waitVm0();
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k4 = 0; k4 < K4H; ++k4) {
          const float4 bv = betaLds[h][k4 * kWave + lane];
          const f32x2 a0 = {a[4 * k4], a[4 * k4 + 1]}, a1 = {a[4 * k4 + 2], a[4 * k4 + 3]};
          const f32x2 b0 = {bv.x, bv.y}, b1 = {bv.z, bv.w};
          const f32x2 q0 = pmul(a0, b0), q1 = pmul(a1, b1);
          w[4 * k4] = q0.x;
          w[4 * k4 + 1] = q0.y;
          w[4 * k4 + 2] = q1.x;
          w[4 * k4 + 3] = q1.y;
        }
        float sumq = 0.f;
        FSMC_W2_ROLE(h, (sumq = w2OrderedTotal<KH, H>(cx, w, kW2RowComb)));
        const float cq = 1.0f / sumq;
        // every read of the landing zone and of this site's ring slot has returned (the barriers above waited for
        // lgkmcnt(0)): request the next site's beta row and the rows of site pos + 2
        if (MODE != kModeSums && pos + 1 < hi) {
          fetchBeta(chunkbuf + (size_t)(pos + 1 - lo) * vecF4);
        }
        if (pos + 2 < stageEnd) {
          stageEmis(pos + 2);
        }

        if (MODE == kModeSums) {
          // HMM::augmentSumOverPairs (HMM.cpp:1052-1081): per site and state, the batch's posteriors are summed over
          // pairs in batch order (local fp32 sum from 0.f).  Every wave transposes the tile of ITS states through its
          // landing zone (64 x 64 floats, row k rotated by k lanes: conflict-free both ways); lane j then owns state j.
          float* const tile = reinterpret_cast<float*>(&betaLds[h][0]);
#pragma unroll
          for (int k = 0; k < KH; ++k) {
            tile[k * kWave + ((lane + k) & (kWave - 1))] = w[k] * cq;
          }
          if (p.flags & FSMC_WANT_MAJOR_MINOR_SUMS) {
            clsLds[h][lane] = (unsigned char)c;
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          waitLgkm0();
          __builtin_amdgcn_wave_barrier();
          const int state = h * KH + lane;
          if (lane < KH && state < K) {
            float s = 0.f, s00 = 0.f, s01 = 0.f, s11 = 0.f;
            for (int v = 0; v < nPairsInGroup; ++v) {
              const float q = tile[lane * kWave + ((v + lane) & (kWave - 1))];
              s = s + q;
              if (p.flags & FSMC_WANT_MAJOR_MINOR_SUMS) {
                const int cv = clsLds[h][v]; // 0 het -> 01, 1 hom major -> 00, 2 hom minor -> 11
                if (cv == 2) {
                  s11 = s11 + q;
                } else if (cv == 1) {
                  s00 = s00 + q;
                } else {
                  s01 = s01 + q;
                }
              }
            }
            float* acc = p.sums + (size_t)blockIdx.x * 4 * p.sumsPlane + (size_t)pos * K + state;
            if (p.flags & FSMC_WANT_SUMS) acc[0] = s;
            if (p.flags & FSMC_WANT_MAJOR_MINOR_SUMS) {
              acc[p.sumsPlane] = s00;
              acc[2 * p.sumsPlane] = s01;
              acc[3 * p.sumsPlane] = s11;
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          waitLgkm0();
          __builtin_amdgcn_wave_barrier();
          if (pos + 1 < hi) {
            fetchBeta(chunkbuf + (size_t)(pos + 1 - lo) * vecF4);
          }
        }

        if (MODE == kModePerPair) {
          // HMM::writePerPairOutput (HMM.cpp:1378-1409): mean = sum_k post*E[t_k] (k ascending from 0.f), MAP = first
          // strictly larger posterior -- both walk the states in order, hence the waves in order (three hand-overs)
          float mean = 0.f, best = 0.f;
          int arg = 0;
#pragma unroll
          for (int ph = 0; ph < kW2NW; ++ph) {
            if (h == ph) {
              if (ph > 0) {
                mean = cx.mail[(kW2RowMean + ph - 1) * kWave + lane];
                best = cx.mail[(kW2RowStep + ph - 1) * kWave + lane];
                arg = __float_as_int(cx.mail[(kW2RowComb + ph - 1) * kWave + lane]);
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
              if (ph < kW2NW - 1) {
                cx.mail[(kW2RowMean + ph) * kWave + lane] = mean;
                cx.mail[(kW2RowStep + ph) * kWave + lane] = best;
                cx.mail[(kW2RowComb + ph) * kWave + lane] = __int_as_float(arg);
              }
            }
            if (ph < kW2NW - 1) {
              w2Barrier();
            }
          }
          if (h == kW2NW - 1 && valid) {
            if (p.ppMean) p.ppMean[(size_t)pairIdx * p.S + pos] = mean;
            if (p.ppMap) p.ppMap[(size_t)pairIdx * p.S + pos] = arg;
          }
        }

        if (MODE == kModeDump) {
          float* out = p.dumpOut + p.dumpOffsets[g] + (size_t)(pos - from) * K * kWave + lane;
#pragma unroll
          for (int k = 0; k < KH; ++k) {
            if (h * KH + k < K) {
              out[(size_t)(h * KH + k) * kWave] = valid ? w[k] * cq : 0.f;
            }
          }
        }

        if (MODE == kModeIbd) {
          if (pos >= scanFrom) {
            // sum over the states below the threshold, k ascending from 0.f (HMM.cpp:1207-1224): wave 0 first, the next
            // waves join only when the threshold reaches their states (uniform over the launch)
            const unsigned nPost = p.stateThr;
            const int nScanWaves = nPost > 3u * KH ? 4 : nPost > 2u * KH ? 3 : nPost > (unsigned)KH ? 2 : 1;
            float s = 0.f;
            auto partial = [&](float s0) -> float {
#pragma unroll
              for (int k4 = 0; k4 < K4H; ++k4) {
                if ((unsigned)(h * KH + 4 * k4) >= nPost) {
                  break;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                  w[4 * k4 + i] = w[4 * k4 + i] * cq;
                  s0 = s0 + ((unsigned)(h * KH + 4 * k4 + i) < nPost ? w[4 * k4 + i] : 0.f);
                }
              }
              return s0;
            };
#pragma unroll
            for (int ph = 0; ph < kW2NW; ++ph) {
              if (ph < nScanWaves) {
                if (h == ph) {
                  s = partial(ph == 0 ? 0.f : cx.mail[(kW2RowScan + ph - 1) * kWave + lane]);
                  if (nScanWaves > 1) {
                    cx.mail[(kW2RowScan + ph) * kWave + lane] = s;
                  }
                }
                if (nScanWaves > 1) {
                  w2Barrier();
                }
              }
            }
            if (nScanWaves > 1) {
              s = cx.mail[(kW2RowScan + nScanWaves - 1) * kWave + lane];
            }
            // the scan's state machine runs in wave 0 (lane = pair)
            int level = 4;
            bool opening = false;
            bool closing = false; // a change of level (or a drop below every threshold) closes the open segment at pos-1
            if (h == 0) {
              level = s >= p.thr[0] ? 0 : s >= p.thr[1] ? 1 : s >= p.thr[2] ? 2 : s >= p.thr[3] ? 3 : 4;
              opening = level != 4 && level != cur;
              closing = valid && cur != 4 && level != cur;
            }
            // the other waves hold states the segment ages read: they need the decision, and wave 0 their sums
            const bool upperAges = TRACK && p.ageThr > (unsigned)KH;
            if (upperAges) {
              const int row = kW2RowLevel + (pos & 1); // (two rows in turn: one barrier a site is enough)
              if (h == 0) {
                cx.mail[row * kWave + lane] = __int_as_float(level | (opening ? 8 : 0) | (closing ? 16 : 0));
              } else {
                // this wave's sums of the sites before are in memory before wave 0 may read them -- but not the requests
                // for the next site's beta row and emission values issued a moment ago, behind those stores (vector
                // memory operations retire in order: "at most that many outstanding" means the stores are done)
                constexpr unsigned nB = (unsigned)K4H, nBE = (unsigned)(K4H + NLE);
                static_assert(nBE < 64, "vmcnt is a 6-bit counter");
                if (pos + 2 < stageEnd) {
                  __builtin_amdgcn_s_waitcnt(0x0F70 | (nBE & 15u) | ((nBE >> 4) << 14));
                } else if (pos + 1 < hi) {
                  __builtin_amdgcn_s_waitcnt(0x0F70 | (nB & 15u) | ((nB >> 4) << 14));
                } else {
                  waitVm0();
                }
              }
              w2Barrier();
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
#pragma unroll
                for (int k4 = 0; k4 < K4H; ++k4) {
                  if ((unsigned)(h * KH + 4 * k4) >= p.ageThr) {
                    break;
                  }
                  const f32x4 t = *rowSlot(spsBase, k4, laneOff);
                  float4 sv = make_float4(t.x, t.y, t.z, t.w);
                  if (opening) {
                    sv = make_float4(0.f, 0.f, 0.f, 0.f);
                  }
                  const float sc = ((unsigned)(h * KH + 4 * k4) < nPost) ? 1.0f : cq;
                  sv.x = sv.x + w[4 * k4] * sc;
                  sv.y = sv.y + w[4 * k4 + 1] * sc;
                  sv.z = sv.z + w[4 * k4 + 2] * sc;
                  sv.w = sv.w + w[4 * k4 + 3] * sc;
                  const f32x4 o = {sv.x, sv.y, sv.z, sv.w};
                  *rowSlot(spsBase, k4, laneOff) = o;
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
              if (h == 0 && valid && cur != 4) {
                emit(segStart, pos);
              }
            }
          }
        }
      }
    }
    // the next group reuses the workspace slot and the mailbox: everything of this one is over in both waves
    waitVm0();
    w2Barrier();
  }
}

} // namespace fsmc
