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
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kMaxGenericK = 256; // upper bound on K for the generic (runtime-K) kernel

enum Mode : int { kModeIbd = 0, kModeDump = 1, kModePerPair = 2, kModeSums = 3 };

struct KParams {
  int K;       // states
  int KP;      // padded row stride of the tables (multiple of 4)
  int S;       // sites
  int W;       // 64-bit words per haplotype row
  int nGroups;
  int chunk;   // sites per chunk (C)
  int chunkRows; // rows of the chunk buffer: C, or (C+1)/2 with beta stride 2
  int maxChunks;
  unsigned flags;
  const float* pi;    // [KP]
  const float* cR;    // [KP]
  const float* expT;  // [KP]
  const float* D;     // [rows][KP]
  const float* B;
  const float* U;
  const float* rowSets; // [rows][5][KP]: D | B | U | Ush | RR of one key side by side, Ush[k] = U[k-1] (packed steps)
  const float* RR;
  const int* stepRow; // [S] row of the step into site q (array mode); sequence mode: the site step, forward
  const int* rowGapF; // sequence mode only: rows of the half-step across the gap (q-1, q), forward
  const int* rowSiteB; //                     site step, backward (out of site q)
  const int* rowGapB;  //                     gap half-step, backward
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
  unsigned long long* phaseCycles; // diagnostic builds (-DFSMC_PHASE_STAMPS): [0] pass B, [1] rebuild, [2] alpha sweep, [3] groups
  float* sums;                // kModeSums: per-slot accumulators [slots][S][K] (+ 00/01/11 planes)
  size_t sumsPlane;           // floats per plane per slot
};

// ---------------------------------------------------------------------------------------------
// Scalar operand streaming.  The k-loops of a step are walked in blocks; each block needs a few
// wave-uniform table values, fetched with hand-placed scalar loads into SGPRs, and this lane's emission
// values (float4 reads from the LDS ring).  The loads of block i+1 are issued right after the wait for
// block i, so that scalar-cache / LDS latency hides under the arithmetic of block i.  Scalar loads are
// inline asm because the compiler otherwise merges and hoists them to the top of the sweep (hundreds
// of spilled SGPRs) -- it cannot see these loads, so each result is only read after an explicit
// s_waitcnt that names it as an in/out operand.  (Waits the compiler inserts for its own LDS reads stay
// correct: extra outstanding scalar loads only make a counted lgkmcnt wait stricter.)
// None of this changes the per-lane order of floating-point operations.
// (hipcc also parses kernel bodies in its host pass, where gfx950 asm constraints do not exist)
#if defined(__HIP_DEVICE_COMPILE__)
#define FSMC_GCN_ASM(...) asm volatile(__VA_ARGS__)
#else
#define FSMC_GCN_ASM(...) ((void)0)
#endif

__device__ __forceinline__ f32x4 sload4(cfloat_p p)
{
  f32x4 v = {};
  FSMC_GCN_ASM("s_load_dwordx4 %0, %1, 0x0" : "=s"(v) : "s"(p));
  return v;
}
__device__ __forceinline__ f32x8 sload8(cfloat_p p)
{
  f32x8 v = {};
  FSMC_GCN_ASM("s_load_dwordx8 %0, %1, 0x0" : "=s"(v) : "s"(p));
  return v;
}
// The same loads with the block's byte offset as an instruction immediate (compile-time K: the unrolled block
// index is a constant by the time the instruction is selected) -- no scalar address arithmetic per load.
__device__ __forceinline__ f32x4 sload4(cfloat_p p, const int byteOff)
{
  f32x4 v = {};
  FSMC_GCN_ASM("s_load_dwordx4 %0, %1, %2" : "=s"(v) : "s"(p), "i"(byteOff));
  return v;
}
__device__ __forceinline__ f32x8 sload8(cfloat_p p, const int byteOff)
{
  f32x8 v = {};
  FSMC_GCN_ASM("s_load_dwordx8 %0, %1, %2" : "=s"(v) : "s"(p), "i"(byteOff));
  return v;
}
// Scalar-cache warm-up.  A table row spans five 64-byte lines; most of them miss the 16-KB scalar cache (16 waves
// share it, each at its own site), and because scalar loads return out of order the operand stream can only wait
// for everything at once -- every block would pay a miss.  touchLines requests one dword of each remaining line of
// a row together with the pass's first operand block: the pass's first wait then covers all the misses in
// parallel and the later blocks hit.  The destination registers are only reserved: they are handed to the next
// wait (swaitTouched) as operands so that nothing else lives in them while the loads are in flight.
#if defined(FSMC_NO_TOUCH)
constexpr bool kTouch = false;
#else
constexpr bool kTouch = true;
#endif
struct Touched {
  float r[4];
};
template <int FIRST, int STEP = 16>
__device__ __forceinline__ void touchLines(Touched& t, cfloat_p p, const int firstState)
{
  // lines FIRST, FIRST+1, ... (up to four) of the row that starts at state firstState; STEP = floats per line
  FSMC_GCN_ASM("s_load_dword %0, %4, %5\n\ts_load_dword %1, %4, %6\n\ts_load_dword %2, %4, %7\n\ts_load_dword %3, %4, %8"
               : "=&s"(t.r[0]), "=&s"(t.r[1]), "=&s"(t.r[2]), "=&s"(t.r[3]) // early clobber: never the address pair
               : "s"(p), "i"((firstState + FIRST * STEP) * 4), "i"((firstState + (FIRST + 1) * STEP) * 4),
                 "i"((firstState + (FIRST + 2) * STEP) * 4), "i"((firstState + (FIRST + 3) * STEP) * 4));
}
__device__ __forceinline__ void holdTouched(Touched& a)
{
  FSMC_GCN_ASM("" : "+s"(a.r[0]), "+s"(a.r[1]), "+s"(a.r[2]), "+s"(a.r[3]));
}
#define FSMC_SWAIT_INSN "s_waitcnt lgkmcnt(0)"
__device__ __forceinline__ f32x16 sload16(cfloat_p p)
{
  f32x16 v = {};
  FSMC_GCN_ASM("s_load_dwordx16 %0, %1, 0x0" : "=s"(v) : "s"(p));
  return v;
}
__device__ __forceinline__ void swait(f32x8& a, f32x8& b)
{
  FSMC_GCN_ASM(FSMC_SWAIT_INSN : "+s"(a), "+s"(b));
}
__device__ __forceinline__ void swait(f32x16& a, f32x16& b)
{
  FSMC_GCN_ASM(FSMC_SWAIT_INSN : "+s"(a), "+s"(b));
}
__device__ __forceinline__ void swait(f32x8& a, f32x8& b, f32x8& c, f32x8& d)
{
  FSMC_GCN_ASM(FSMC_SWAIT_INSN : "+s"(a), "+s"(b), "+s"(c), "+s"(d));
}
// block-width dispatch so the block sizes below are tunable
template <int N> struct SV;
template <> struct SV<4> {
  typedef f32x4 T;
  static __device__ __forceinline__ T load(cfloat_p p) { return sload4(p); }
  static __device__ __forceinline__ T loadAt(cfloat_p p, const int firstState) { return sload4(p, firstState * 4); }
};
template <> struct SV<8> {
  typedef f32x8 T;
  static __device__ __forceinline__ T load(cfloat_p p) { return sload8(p); }
  static __device__ __forceinline__ T loadAt(cfloat_p p, const int firstState) { return sload8(p, firstState * 4); }
};
template <> struct SV<16> {
  typedef f32x16 T;
  static __device__ __forceinline__ T load(cfloat_p p) { return sload16(p); }
  static __device__ __forceinline__ T loadAt(cfloat_p p, const int firstState)
  {
    f32x16 v = {};
    FSMC_GCN_ASM("s_load_dwordx16 %0, %1, %2" : "=s"(v) : "s"(p), "i"(firstState * 4));
    return v;
  }
};
__device__ __forceinline__ void swait(f32x4& a, f32x4& b, f32x4& c, f32x4& d)
{
  FSMC_GCN_ASM(FSMC_SWAIT_INSN : "+s"(a), "+s"(b), "+s"(c), "+s"(d));
}

#if defined(FSMC_PHASE_STAMPS)
#define FSMC_SWAIT(acc, ...)                                                                                           \
  do {                                                                                                                 \
    const long long t0_ = (long long)clock64();                                                                        \
    swait(__VA_ARGS__);                                                                                                \
    (acc) += (long long)clock64() - t0_;                                                                               \
  } while (0)
#else
#define FSMC_SWAIT(acc, ...) swait(__VA_ARGS__)
#endif

#ifndef FSMC_KB
#define FSMC_KB 16
#endif
#ifndef FSMC_KBF
#define FSMC_KBF 8
#endif
constexpr int kKB = FSMC_KB;   // states per operand block of the beta passes (two tables at a time)
constexpr int kKBF = FSMC_KBF; // states per operand block of the alpha pass (four tables at a time)
constexpr int kKPad = 16;      // table / emission rows are zero padded to a multiple of this many floats
// RowSet: the transition-table rows of one key side by side, [key][5][KP] floats (packed steps)
enum RowSetPart : int { kRowD = 0, kRowB = 1, kRowU = 2, kRowUsh = 3, kRowRR = 4, kRowSetParts = 5 };

__device__ __forceinline__ float pick(const float4& e0, const float4& e1, const int i)
{
  return i == 0 ? e0.x : i == 1 ? e0.y : i == 2 ? e0.z : i == 3 ? e0.w : i == 4 ? e1.x : i == 5 ? e1.y : i == 6 ? e1.z : e1.w;
}

template <int N> struct EmisBlk { // this lane's emission values of one operand block (N/4 float4 from LDS)
  float4 v[N / 4];
  __device__ __forceinline__ float at(const int i) const
  {
    const float4& q = v[i >> 2];
    return (i & 3) == 0 ? q.x : (i & 3) == 1 ? q.y : (i & 3) == 2 ? q.z : q.w;
  }
  __device__ __forceinline__ f32x2 pair(const int i) const // values i, i+1 (i even): one 64-bit register pair
  {
    const float4& q = v[i >> 2];
    const f32x2 lo = {q.x, q.y}, hi = {q.z, q.w};
    return (i & 2) == 0 ? lo : hi;
  }
};
// values i, i+1 (i even) of a scalar operand block: an aligned SGPR pair
template <typename V> __device__ __forceinline__ f32x2 pairOf(const V& v, const int i)
{
  const f32x2 r = {v[i], v[i + 1]};
  return r;
}
template <int N> __device__ __forceinline__ EmisBlk<N> readEmis(const float4* e, const int blk)
{
  EmisBlk<N> r;
#pragma unroll
  for (int j = 0; j < N / 4; ++j) {
    r.v[j] = e[blk * (N / 4) + j];
  }
  return r;
}

// One step of the backward recursion for one pair (HMM.cpp:957-1016, NO_SSE association).
// b: beta of site pos+1 (scaled) on entry, beta of site pos (scaled) on exit.  w: scratch.
// e: this lane's emission row for site pos+1 (LDS).  Dr/Br/Ur/RRr: wave-uniform table rows.
// SCALE = false: the un-normalised half-step of sequence mode (HMM.cpp:915-922).
template <int KT, int KA, bool SCALE = true>
__device__ __forceinline__ void beta_step_1(const int K, float (&b)[KA], float (&w)[KA], cfloat_p Dr, cfloat_p Br,
                                            cfloat_p Ur, cfloat_p RRr, const float4* e, long long& waitCycles)
{
  typedef typename SV<kKB>::T SVec;
  const int NB = (K + kKB - 1) / kKB;
  // descending: vec[k] = beta[k]*e[k];  BU[k] = U[k]*vec[k+1] + RR[k]*BU[k+1]  (BU[K-1] = 0)
  SVec u = SV<kKB>::load(Ur + (NB - 1) * kKB);
  SVec rr = SV<kKB>::load(RRr + (NB - 1) * kKB);
  EmisBlk<kKB> em = readEmis<kKB>(e, NB - 1);
  SVec d, bt; // operands of the ascending pass; its first block is requested during the last descending block
  // With a runtime K these loops are real loops: a register holding a scalar load still in flight must not be
  // copied across the back-edge, so the generic instantiation loads and waits per block instead of prefetching.
#pragma unroll
  for (int blk = NB - 1; blk >= 0; --blk) {
    if (KT == 0 && blk < NB - 1) {
      u = SV<kKB>::load(Ur + blk * kKB);
      rr = SV<kKB>::load(RRr + blk * kKB);
      em = readEmis<kKB>(e, blk);
    }
    FSMC_SWAIT(waitCycles, u, rr);
    SVec nu = u, nrr = rr;
    EmisBlk<kKB> nem = em;
    if (KT > 0) {
      if (blk > 0) {
        nu = SV<kKB>::load(Ur + (blk - 1) * kKB);
        nrr = SV<kKB>::load(RRr + (blk - 1) * kKB);
        nem = readEmis<kKB>(e, blk - 1);
      } else {
        d = SV<kKB>::load(Dr);
        bt = SV<kKB>::load(Br);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = kKB - 1; i >= 0; --i) {
      const int k = blk * kKB + i;
      if (k < K) {
        b[k] = b[k] * em.at(i);
        if (k == K - 1) {
          w[k] = 0.f;
        } else {
          w[k] = u[i] * b[k + 1] + rr[i] * w[k + 1];
        }
      }
    }
    u = nu;
    rr = nrr;
    em = nem;
  }
  // ascending: BL[k] = BL[k-1] + B[k-1]*vec[k-1];  beta'[k] = (BL[k] + D[k]*vec[k]) + BU[k]
  float BL = 0.f;
  float sum = 0.f;
#pragma unroll
  for (int blk = 0; blk < NB; ++blk) {
    if (KT == 0) {
      d = SV<kKB>::load(Dr + blk * kKB);
      bt = SV<kKB>::load(Br + blk * kKB);
    }
    FSMC_SWAIT(waitCycles, d, bt);
    SVec nd = d, nbt = bt;
    if (KT > 0 && blk + 1 < NB) {
      nd = SV<kKB>::load(Dr + (blk + 1) * kKB);
      nbt = SV<kKB>::load(Br + (blk + 1) * kKB);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < kKB; ++i) {
      const int k = blk * kKB + i;
      if (k < K) {
        w[k] = (BL + d[i] * b[k]) + w[k];
        sum = sum + w[k];
        if (k < K - 1) {
          BL = BL + bt[i] * b[k];
        }
      }
    }
    d = nd;
    bt = nbt;
  }
  if constexpr (SCALE) {
    const float c = 1.0f / sum;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      b[k] = w[k] * c;
    }
  } else {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      b[k] = w[k];
    }
  }
}

// One step of the forward recursion (HMM.cpp:799-830) followed by the per-site scaling
// (HmmUtils.cpp:102-151).  a: alpha of site pos-1 on entry, of site pos on exit.
// SCALE = false: the un-normalised half-step of sequence mode (HMM.cpp:760-767).
template <int KT, int KA, bool SCALE = true>
__device__ __forceinline__ void alpha_step_1(const int K, float (&a)[KA], float (&w)[KA], cfloat_p Dr, cfloat_p Br,
                                             cfloat_p Ur, cfloat_p cR, const float4* e, long long& waitCycles)
{
  typedef typename SV<kKBF>::T SVec;
  const int NB = (K + kKBF - 1) / kKBF;
  // first operand block requested before the operand-free suffix-sum pass
  SVec d = SV<kKBF>::load(Dr), bt = SV<kKBF>::load(Br), u = SV<kKBF>::load(Ur), c4 = SV<kKBF>::load(cR);
  EmisBlk<kKBF> em = readEmis<kKBF>(e, 0);
  __builtin_amdgcn_sched_barrier(0);
  // alphaC[k] = sum_{i>=k} alpha[i], accumulated from the top (HMM.cpp:799-814)
  w[K - 1] = a[K - 1];
#pragma unroll
  for (int k = K - 2; k >= 0; --k) {
    w[k] = w[k + 1] + a[k];
  }
  float AU = 0.f;
  float sum = 0.f;
#pragma unroll
  for (int blk = 0; blk < NB; ++blk) {
    if (KT == 0 && blk > 0) {
      d = SV<kKBF>::load(Dr + blk * kKBF);
      bt = SV<kKBF>::load(Br + blk * kKBF);
      u = SV<kKBF>::load(Ur + blk * kKBF);
      c4 = SV<kKBF>::load(cR + blk * kKBF);
      em = readEmis<kKBF>(e, blk);
    }
    FSMC_SWAIT(waitCycles, d, bt, u, c4);
    SVec nd = d, nbt = bt, nu = u, nc = c4;
    EmisBlk<kKBF> nem = em;
    if (KT > 0 && blk + 1 < NB) {
      nd = SV<kKBF>::load(Dr + (blk + 1) * kKBF);
      nbt = SV<kKBF>::load(Br + (blk + 1) * kKBF);
      nu = SV<kKBF>::load(Ur + (blk + 1) * kKBF);
      nc = SV<kKBF>::load(cR + (blk + 1) * kKBF);
      nem = readEmis<kKBF>(e, blk + 1);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < kKBF; ++i) {
      const int k = blk * kKBF + i;
      if (k < K) {
        float term = AU + d[i] * a[k];
        if (k < K - 1) {
          term = term + bt[i] * w[k + 1];
        }
        w[k] = em.at(i) * term;
        sum = sum + w[k];
        if (k < K - 1) {
          AU = u[i] * a[k] + c4[i] * AU; // AU of state k+1
        }
      }
    }
    d = nd;
    bt = nbt;
    u = nu;
    c4 = nc;
    em = nem;
  }
  if constexpr (SCALE) {
    const float c = 1.0f / sum;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      a[k] = w[k] * c;
    }
  } else {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      a[k] = w[k];
    }
  }
}


// ---------------------------------------------------------------------------------------------
// Packed variants of the two steps for a compile-time K.  gfx950 multiplies / adds two fp32 values per lane in one
// VALU instruction (v_pk_mul_f32, v_pk_add_f32) when both sit in an aligned register pair.  Every operation of a
// step that is not part of a first-order recurrence is done for states (k, k+1) at once; the recurrences (BU, BL,
// AU, the suffix sum, the scaling sum) stay scalar and sequential.  Each value is produced by the same IEEE
// operation on the same operands as in the scalar step (no FMA, no re-association), so the results are
// bit-identical; only ~7.5 instead of 11 VALU instructions are issued per state.
//
// Backward: the term U[k]*vec[k+1] of BU[k] is taken from T[m] = Ush[m]*vec[m] with Ush[m] = U[m-1]
// (a second copy of the U table shifted by one state), so that both factors share a state index.
// First operand block of a backward step, requested by the step before it (BetaHead): the descending pass opens
// with a wait on operands that nothing can be overlapped with unless they were asked for during the previous step's
// last block.  The previous step waits for them before it returns (after its scaling loop), so the values are final
// when they cross the code between two steps.
struct BetaHead {
  typename SV<kKB>::T u, rr;
};

template <int KT, int KA, bool SCALE = true>
__device__ __forceinline__ void beta_step_pk(float (&b)[KA], float (&w)[KA], cfloat_p rowSet, const float4* e,
                                             long long& waitCycles, BetaHead& head, cfloat_p nextRowSet)
{
  constexpr int K = KT;
  // the five table rows of one key sit side by side (RowSet): one base register, block offsets as immediates
  constexpr int KPc = ((KT + kKPad - 1) / kKPad) * kKPad;
  typedef typename SV<kKB>::T SVec;
  constexpr int NB = (K + kKB - 1) / kKB; // operand blocks (scalar loads, one block ahead)
  constexpr int R = kKB / 8;              // emission sub-blocks of 8 states per operand block
  constexpr int NSB = (K + 7) / 8;
  static_assert(!kTouch || (KPc == 80 && K > 64), "line warm-up is laid out for rows of five 64-byte lines");
  SVec u = head.u, rr = head.rr; // requested and waited for by betaHeadPrime or by the previous step
  Touched tu, trr, td, tbt;
  SVec nu = u, nrr = rr;
  EmisBlk<8> em = readEmis<8>(e, NSB - 1);
  SVec d, bt;
  float tcarry = 0.f; // T of the first state of the sub-block above
#pragma unroll
  for (int sb = NSB - 1; sb >= 0; --sb) {
    const int blk = sb / R;
    const int o = (sb % R) * 8; // offset of this sub-block inside the operand block
    if (sb == NSB - 1 || sb % R == R - 1) {
      // entering operand block blk: its loads were issued a whole block ago; request the next one
      if (sb != NSB - 1) {
        FSMC_SWAIT(waitCycles, nu, nrr);
        u = nu;
        rr = nrr;
      }
      if (blk > 0) {
        nu = SV<kKB>::loadAt(rowSet, kRowUsh * KPc + (blk - 1) * kKB);
        nrr = SV<kKB>::loadAt(rowSet, kRowRR * KPc + (blk - 1) * kKB);
      } else {
        d = SV<kKB>::loadAt(rowSet, kRowD * KPc);
        bt = SV<kKB>::loadAt(rowSet, kRowB * KPc);
        if constexpr (kTouch) {
          touchLines<1>(td, rowSet, kRowD * KPc);
          touchLines<1>(tbt, rowSet, kRowB * KPc);
        }
      }
    }
    EmisBlk<8> nem = em;
    if (sb > 0) {
      nem = readEmis<8>(e, sb - 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    float T[9];
    T[8] = tcarry;
    // vec[k] = beta[k]*e[k] (kept in b), T[k] = U[k-1]*vec[k]
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
      const int k = sb * 8 + i;
      if (k + 1 < K) {
        f32x2 v = {b[k], b[k + 1]};
        v = v * em.pair(i);
        const f32x2 t = pairOf(u, o + i) * v;
        b[k] = v.x;
        b[k + 1] = v.y;
        T[i] = t.x;
        T[i + 1] = t.y;
      } else if (k < K) {
        b[k] = b[k] * em.at(i);
        T[i] = u[o + i] * b[k];
      }
    }
    // BU[k] = U[k]*vec[k+1] + RR[k]*BU[k+1], BU[K-1] = 0 (HMM.cpp:986-1005)
#pragma unroll
    for (int i = 7; i >= 0; --i) {
      const int k = sb * 8 + i;
      if (k < K) {
        if (k == K - 1) {
          w[k] = 0.f;
        } else {
          w[k] = T[i + 1] + rr[o + i] * w[k + 1];
        }
      }
    }
    tcarry = T[0];
    em = nem;
  }
  // ascending: BL[k] = BL[k-1] + B[k-1]*vec[k-1];  beta'[k] = (BL[k] + D[k]*vec[k]) + BU[k]
  float BL = 0.f;
  float sum = 0.f;
  SVec nd = d, nbt = bt;
#pragma unroll
  for (int blk = 0; blk < NB; ++blk) {
    if (blk > 0) {
      FSMC_SWAIT(waitCycles, nd, nbt);
      d = nd;
      bt = nbt;
    } else {
      FSMC_SWAIT(waitCycles, d, bt);
      if constexpr (kTouch) {
        holdTouched(td);
        holdTouched(tbt);
      }
    }
    if (blk + 1 < NB) {
      nd = SV<kKB>::loadAt(rowSet, kRowD * KPc + (blk + 1) * kKB);
      nbt = SV<kKB>::loadAt(rowSet, kRowB * KPc + (blk + 1) * kKB);
    } else {
      // last block: the next step's first operand block and the lines of its descending pass (always requested --
      // a branch here would split the step's single basic block; without a next step the caller passes this row)
      head.u = SV<kKB>::loadAt(nextRowSet, kRowUsh * KPc + (NB - 1) * kKB);
      head.rr = SV<kKB>::loadAt(nextRowSet, kRowRR * KPc + (NB - 1) * kKB);
      if constexpr (kTouch) {
        touchLines<0>(tu, nextRowSet, kRowUsh * KPc);
        touchLines<0>(trr, nextRowSet, kRowRR * KPc);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < kKB; i += 2) {
      const int k = blk * kKB + i;
      if (k + 1 < K) {
        const f32x2 v = {b[k], b[k + 1]};
        const f32x2 dv = pairOf(d, i) * v;
        const f32x2 bv = pairOf(bt, i) * v;
        f32x2 bl;
        bl.x = BL;
        bl.y = BL + bv.x;
        f32x2 x = bl + dv;
        const f32x2 bu = {w[k], w[k + 1]};
        x = x + bu;
        w[k] = x.x;
        w[k + 1] = x.y;
        sum = sum + x.x;
        sum = sum + x.y;
        BL = (k + 1 < K - 1) ? bl.y + bv.y : bl.y;
      } else if (k < K) {
        w[k] = (BL + d[i] * b[k]) + w[k];
        sum = sum + w[k];
        if (k < K - 1) {
          BL = BL + bt[i] * b[k];
        }
      }
    }
  }
  if constexpr (SCALE) {
    const float c = 1.0f / sum;
    const f32x2 cc = {c, c};
#pragma unroll
    for (int k = 0; k < K; k += 2) {
      if (k + 1 < K) {
        const f32x2 x = {w[k], w[k + 1]};
        const f32x2 y = x * cc;
        b[k] = y.x;
        b[k + 1] = y.y;
      } else {
        b[k] = w[k] * c;
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      b[k] = w[k];
    }
  }
  FSMC_SWAIT(waitCycles, head.u, head.rr);
  if constexpr (kTouch) {
    holdTouched(tu);
    holdTouched(trr);
  }
}

// The first operand block of a backward step that no step precedes (start of a sweep, a recomputed row).
template <int KT> __device__ __forceinline__ void betaHeadPrime(BetaHead& head, cfloat_p rowSet, long long& waitCycles)
{
  constexpr int KPc = ((KT + kKPad - 1) / kKPad) * kKPad;
  constexpr int NB = (KT + kKB - 1) / kKB;
  head.u = SV<kKB>::loadAt(rowSet, kRowUsh * KPc + (NB - 1) * kKB);
  head.rr = SV<kKB>::loadAt(rowSet, kRowRR * KPc + (NB - 1) * kKB);
  Touched tu, trr;
  if constexpr (kTouch) {
    touchLines<0>(tu, rowSet, kRowUsh * KPc);
    touchLines<0>(trr, rowSet, kRowRR * KPc);
  }
  FSMC_SWAIT(waitCycles, head.u, head.rr);
  if constexpr (kTouch) {
    holdTouched(tu);
    holdTouched(trr);
  }
}

// Forward.  The suffix sums are kept one slot down (w[k] = alphaC[k+1]) so that B[k]*alphaC[k+1] pairs up with the
// other products of state k; alphaC[0] is never used (HMM.cpp:799-830).
template <int KT, int KA, bool SCALE = true>
__device__ __forceinline__ void alpha_step_pk(float (&a)[KA], float (&w)[KA], cfloat_p rowSet, cfloat_p cR,
                                              const float4* e, long long& waitCycles)
{
  constexpr int K = KT;
  constexpr int KPc = ((KT + kKPad - 1) / kKPad) * kKPad;
  static_assert(K >= 2, "packed step needs at least two states");
  typedef typename SV<kKBF>::T SVec;
  constexpr int NB = (K + kKBF - 1) / kKBF; // operand blocks
  constexpr int R = kKBF / 4;               // emission sub-blocks of 4 states per operand block
  constexpr int NSB = (K + 3) / 4;
  SVec d = SV<kKBF>::loadAt(rowSet, kRowD * KPc), bt = SV<kKBF>::loadAt(rowSet, kRowB * KPc),
       u = SV<kKBF>::loadAt(rowSet, kRowU * KPc), c4 = SV<kKBF>::loadAt(cR, 0);
  Touched td, tbt, tu;
  if constexpr (kTouch) {
    touchLines<1>(td, rowSet, kRowD * KPc);
    touchLines<1>(tbt, rowSet, kRowB * KPc);
    touchLines<1>(tu, rowSet, kRowU * KPc);
  }
  SVec nd = d, nbt = bt, nu = u, nc = c4;
  EmisBlk<4> em = readEmis<4>(e, 0);
  __builtin_amdgcn_sched_barrier(0);
  w[K - 2] = a[K - 1];
#pragma unroll
  for (int k = K - 2; k >= 1; --k) {
    w[k - 1] = w[k] + a[k];
  }
  float AU = 0.f;
  float sum = 0.f;
#pragma unroll
  for (int sb = 0; sb < NSB; ++sb) {
    const int blk = sb / R;
    const int o = (sb % R) * 4;
    if (sb % R == 0) {
      if (sb > 0) {
        FSMC_SWAIT(waitCycles, nd, nbt, nu, nc);
        d = nd;
        bt = nbt;
        u = nu;
        c4 = nc;
      } else {
        FSMC_SWAIT(waitCycles, d, bt, u, c4);
        if constexpr (kTouch) {
          holdTouched(td);
          holdTouched(tbt);
          holdTouched(tu);
        }
      }
      if (blk + 1 < NB) {
        nd = SV<kKBF>::loadAt(rowSet, kRowD * KPc + (blk + 1) * kKBF);
        nbt = SV<kKBF>::loadAt(rowSet, kRowB * KPc + (blk + 1) * kKBF);
        nu = SV<kKBF>::loadAt(rowSet, kRowU * KPc + (blk + 1) * kKBF);
        nc = SV<kKBF>::loadAt(cR, (blk + 1) * kKBF);
      }
    }
    EmisBlk<4> nem = em;
    if (sb + 1 < NSB) {
      nem = readEmis<4>(e, sb + 1);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; i += 2) {
      const int k = sb * 4 + i;
      if (k + 1 < K - 1) {
        const f32x2 av = {a[k], a[k + 1]};
        const f32x2 da = pairOf(d, o + i) * av;
        const f32x2 ua = pairOf(u, o + i) * av;
        const f32x2 ac = {w[k], w[k + 1]};
        const f32x2 bw = pairOf(bt, o + i) * ac;
        f32x2 au;
        au.x = AU;
        au.y = ua.x + c4[o + i] * AU; // AU of state k+1
        f32x2 term = au + da;
        term = term + bw;
        const f32x2 ov = em.pair(i) * term;
        w[k] = ov.x;
        w[k + 1] = ov.y;
        sum = sum + ov.x;
        sum = sum + ov.y;
        AU = ua.y + c4[o + i + 1] * au.y; // AU of state k+2
      } else {
#pragma unroll
        for (int ii = i; ii < i + 2; ++ii) {
          const int kk = sb * 4 + ii;
          if (kk < K) {
            float term = AU + d[o + ii] * a[kk];
            if (kk < K - 1) {
              term = term + bt[o + ii] * w[kk];
            }
            w[kk] = em.at(ii) * term;
            sum = sum + w[kk];
            if (kk < K - 1) {
              AU = u[o + ii] * a[kk] + c4[o + ii] * AU;
            }
          }
        }
      }
    }
    em = nem;
  }
  if constexpr (SCALE) {
    const float c = 1.0f / sum;
    const f32x2 cc = {c, c};
#pragma unroll
    for (int k = 0; k < K; k += 2) {
      if (k + 1 < K) {
        const f32x2 x = {w[k], w[k + 1]};
        const f32x2 y = x * cc;
        a[k] = y.x;
        a[k + 1] = y.y;
      } else {
        a[k] = w[k] * c;
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      a[k] = w[k];
    }
  }
}



#if defined(FSMC_NO_PK)
constexpr bool kPacked = false;
#else
constexpr bool kPacked = true;
#endif

// The tables as the kernel addresses them: `row` selects the key.  Packed steps read the RowSet copy, the generic
// (runtime-K) steps the four separate tables.
struct Tables {
  cfloat_p D, B, U, RR, rowSets, cR;
  int KP;
};

// nextRow: key of the backward step that follows immediately (its first operands are requested ahead through
// `head`), or -1.
template <int KT, int KA, bool SCALE = true>
__device__ __forceinline__ void beta_step(const int K, float (&b)[KA], float (&w)[KA], const Tables& t, const int row,
                                          const float4* e, long long& waitCycles, BetaHead& head,
                                          const bool primed = false, const int nextRow = -1)
{
  if constexpr (KT > 0 && kPacked) {
    constexpr int KPc = ((KT + kKPad - 1) / kKPad) * kKPad;
    const cfloat_p rs = t.rowSets + (size_t)row * (kRowSetParts * KPc);
#if defined(FSMC_NO_HEAD)
    betaHeadPrime<KT>(head, rs, waitCycles);
    beta_step_pk<KT, KA, SCALE>(b, w, rs, e, waitCycles, head, rs);
#else
    if (!primed) {
      betaHeadPrime<KT>(head, rs, waitCycles);
    }
    beta_step_pk<KT, KA, SCALE>(b, w, rs, e, waitCycles, head,
                                t.rowSets + (size_t)(nextRow < 0 ? row : nextRow) * (kRowSetParts * KPc));
#endif
  } else {
    const size_t o = (size_t)row * t.KP;
    beta_step_1<KT, KA, SCALE>(K, b, w, t.D + o, t.B + o, t.U + o, t.RR + o, e, waitCycles);
  }
}

template <int KT, int KA, bool SCALE = true>
__device__ __forceinline__ void alpha_step(const int K, float (&a)[KA], float (&w)[KA], const Tables& t, const int row,
                                           const float4* e, long long& waitCycles)
{
  if constexpr (KT > 0 && kPacked) {
    constexpr int KPc = ((KT + kKPad - 1) / kKPad) * kKPad;
    alpha_step_pk<KT, KA, SCALE>(a, w, t.rowSets + (size_t)row * (kRowSetParts * KPc), t.cR, e, waitCycles);
  } else {
    const size_t o = (size_t)row * t.KP;
    alpha_step_1<KT, KA, SCALE>(K, a, w, t.D + o, t.B + o, t.U + o, t.cR, e, waitCycles);
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
    // streamed once, read back once: keep it from evicting the model tables out of L2
    const f32x4 ov = {o.x, o.y, o.z, o.w};
    __builtin_nontemporal_store(ov, reinterpret_cast<f32x4*>(&dst[(size_t)k4 * kWave]));
  }
}

template <int KT, int KA> __device__ __forceinline__ void load_vec(const int K, const float4* src, float (&v)[KA])
{
  const int K4 = (K + 3) >> 2;
#pragma unroll
  for (int k4 = 0; k4 < K4; ++k4) {
    const f32x4 ov = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(&src[(size_t)k4 * kWave]));
    const float4 o = make_float4(ov.x, ov.y, ov.z, ov.w);
    v[4 * k4] = o.x;
    if (4 * k4 + 1 < K) v[4 * k4 + 1] = o.y;
    if (4 * k4 + 2 < K) v[4 * k4 + 2] = o.z;
    if (4 * k4 + 3 < K) v[4 * k4 + 3] = o.w;
  }
}

// Segment age estimates from the per-state posterior sums of a segment
// (HMM::getPosteriorMean, HMM.cpp:1087-1097; HMM::getMAP, 1099-1107).  The sums live in the wave's workspace
// as [K/4][64 lanes] float4 (sps points at this lane's column); this runs once per IBD record, so it walks
// memory with real loops instead of holding a K-vector in registers.
__device__ __forceinline__ float spsAt(const float4* sps, const int k)
{
  return reinterpret_cast<const float*>(sps + (size_t)(k >> 2) * kWave)[k & 3];
}
__device__ __forceinline__ void segment_ages(const int K, const unsigned nAge, const float4* sps, cfloat_p pi,
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
      acc = acc + spsAt(sps, k);
    }
    const float norm = 1.f / acc;
#pragma nounroll
    for (int k = 0; k < n; ++k) {
      mean = mean + (norm * spsAt(sps, k)) * expT[k];
    }
  }
  if (wantMap) {
    float best = 0.f;
    float bestT = 0.f;
#pragma nounroll
    for (int k = 0; k < n; ++k) {
      const float r = spsAt(sps, k) / pi[k];
      if (k == 0 || best < r) {
        best = r;
        bestT = expT[k];
      }
    }
    mapv = bestT;
  }
}

// SEQ: sequence mode (DecodingParams::decodingSequence) -- every site step is preceded by an un-normalised
// half-step across the homozygous stretch since the neighbouring site, and the vectors the posterior is built
// from are the ones the reference's buffers end up holding (HMM.cpp:767, 922; oracle/hmm_oracle.h):
//   stored beta of site p  = beta after the half-step towards p-1 (p > from),
//   stored alpha of site p = alpha after the half-step towards p+1 (p < to-1).
// The emission ring then carries a fourth row per site: the homozygous emission of the gap before it.
//
// HALF: beta stride 2.  Within a chunk [lo, hi) only the beta rows of the sites at odd offsets
// (and of the chunk's last site) are written to HBM; the alpha sweep recomputes the row of an even-offset site
// from its successor's row (already landed in LDS) with one more beta step.  Half the HBM traffic of the beta
// stream for half a sweep of extra arithmetic; the floating-point operations of every row are unchanged.
template <int KT, int MODE, bool TRACK, bool SEQ, bool HALF>
__global__ __launch_bounds__(kWave, 2) void decode_kernel(const KParams p)
{
  static_assert(!HALF || (!SEQ && MODE == kModeIbd), "beta stride 2 is built for the array-mode IBD decode");
  constexpr int KA = KT > 0 ? KT : kMaxGenericK;
  constexpr int K4A = (KA + 3) / 4;
  constexpr int E4A = ((KA + kKPad - 1) / kKPad) * (kKPad / 4); // float4 per emission row (rows padded to kKPad)
  constexpr int NC = SEQ ? 4 : 3;                         // emission rows per site: 3 observation classes (+ gap)
  constexpr int NL = (NC * E4A + kWave - 1) / kWave;      // float4 per lane to stage one site's rows
  const int K = KT > 0 ? KT : p.K;
  const int K4 = (K + 3) >> 2;
  const int KP = p.KP; // (a compile-time KP for fixed K measured 6 % slower on C2: keep the runtime value)
  const int E4 = KP >> 2;

  __shared__ float4 emisLds[2][NC * E4A]; // ring of two sites x three observation classes (+ the gap row)
  __shared__ float4 betaLds[K4A * kWave]; // landing zone of the next site's beta row (LDS-DMA)

  const int lane = threadIdx.x;
  const cfloat_p tD = (cfloat_p)p.D, tB = (cfloat_p)p.B, tU = (cfloat_p)p.U, tRR = (cfloat_p)p.RR;
  const cfloat_p tPi = (cfloat_p)p.pi, tCR = (cfloat_p)p.cR, tExpT = (cfloat_p)p.expT;
  const Tables tabs = {tD, tB, tU, tRR, (cfloat_p)p.rowSets, tCR, KP};
  const cint_p tStepRow = (cint_p)p.stepRow;
  const cint_p tRowGapF = (cint_p)p.rowGapF, tRowSiteB = (cint_p)(SEQ ? p.rowSiteB : p.stepRow),
               tRowGapB = (cint_p)p.rowGapB;
  const size_t vecF4 = (size_t)K4 * kWave; // float4 per stored K-vector of a wave
  float4* const chunkbuf = p.ws + (size_t)blockIdx.x * p.wsSlot;
  float4* const ckpt = chunkbuf + (size_t)p.chunkRows * vecF4;
  float4* const saveA = ckpt + (size_t)(p.maxChunks + 2) * vecF4;
  float4* const saveS = saveA + vecF4;
  const int C = p.chunk;
  // per-state posterior sums of the open segments (TRACK), one column per lane
  float4* const spsMem = saveS + threadIdx.x;
  // chunk-buffer slot of the row stored for the site at offset rel of its chunk
  auto slotOf = [](const int rel) -> size_t { return (size_t)(HALF ? (rel >> 1) : rel); };

  struct EmisRegs {
    float4 v[NL];
  };

  for (unsigned round = 0;; ++round) {
    unsigned g = 0;
    if (MODE == kModeSums) {
      // each resident wave owns an accumulator plane and takes groups slot, slot + nSlots, ... in order, so the
      // order of float additions is fixed from run to run
      g = blockIdx.x + round * gridDim.x;
    } else {
      if (lane == 0) {
        g = atomicAdd(&p.counters[0], 1u);
      }
      g = __builtin_amdgcn_readfirstlane(g);
    }
    if (g >= (unsigned)p.nGroups) {
      break;
    }
    const cuint_p gw = (cuint_p)(p.groups + g);
    const unsigned firstPair = gw[0];
    const int nPairsInGroup = (int)gw[1];
    const int from = (int)gw[2];
    const int to = (int)gw[3];
    const int scanFrom = (int)gw[4];
    const int aEnd = (MODE == kModeIbd) ? (int)gw[5] : to; // the alpha sweep stops here
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
    auto prefetchEmis = [&](const int q) -> EmisRegs {
      EmisRegs r;
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const int idx = lane + i * kWave;
        r.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < NC * E4) {
          r.v[i] = p.emis3[(size_t)q * (NC * E4) + idx];
        }
      }
      return r;
    };
    auto commitEmis = [&](const int q, const EmisRegs& r) {
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const int idx = lane + i * kWave;
        if (idx < NC * E4) {
          emisLds[q & 1][idx] = r.v[i];
        }
      }
      __builtin_amdgcn_wave_barrier();
    };

    float w[KA];
    long long cycW = 0; // cycles parked in operand waits (diagnostic builds only)
#if defined(FSMC_PHASE_STAMPS)
    // diagnostic build only: where a group's wall time goes (never enabled in the shipped library)
    long long cycB = 0, cycR = 0, cycA = 0;
    long long stamp = (long long)clock64();
#define FSMC_STAMP(acc)                                                                                                \
  do {                                                                                                                 \
    const long long now_ = (long long)clock64();                                                                       \
    (acc) += now_ - stamp;                                                                                             \
    stamp = now_;                                                                                                      \
  } while (0)
#else
#define FSMC_STAMP(acc) ((void)0)
#endif

    // Sequence mode, backward.  The vector carried from site to site is the STORED one (after the half-step).
    // betaGapStep: stage site q's rows (its fourth row is the homozygous emission of the gap (q-1, q)) and take
    // the un-normalised half-step across that gap.  betaSeqStep: the site step out of q = pos+1 (whose rows the
    // previous half-step left in the ring), then the half-step towards pos-1 unless pos is the window start.
    auto betaGapStep = [&](float (&b)[KA], const int q, const EmisRegs& rows) {
      commitEmis(q, rows);
      const int row = tRowGapB[q];
      BetaHead head;
      beta_step<KT, KA, false>(K, b, w, tabs, row, &emisLds[q & 1][3 * E4], cycW, head);
    };
    auto betaSeqStep = [&](float (&b)[KA], const int pos) {
      const int q = pos + 1;
      const bool gap = pos > from;
      EmisRegs ev;
      if (gap) {
        ev = prefetchEmis(pos);
      }
      const int c = obsClass(q);
      const int row = tRowSiteB[q];
      BetaHead head;
      beta_step<KT, KA>(K, b, w, tabs, row, &emisLds[q & 1][c * E4], cycW, head);
      if (gap) {
        betaGapStep(b, pos, ev);
      }
    };

    // ------------------------------------------------------------------ pass B
    {
      float b[KA];
      beta_init<KT, KA>(K, b);
      auto afterBeta = [&](const int pos) {
        if (single) {
          const int rel = pos - from;
          if (pos < aEnd && (!HALF || (rel & 1) || pos == aEnd - 1)) {
            store_vec<KT, KA>(K, chunkbuf + slotOf(rel) * vecF4 + lane, b);
          }
        } else {
          const int rel = pos - from;
          if (rel > 0 && pos <= aEnd && (rel % C == 0 || pos == aEnd)) {
            const int j = (pos == aEnd) ? nChunks : rel / C;
            store_vec<KT, KA>(K, ckpt + (size_t)j * vecF4 + lane, b);
          }
        }
      };
      if constexpr (SEQ) {
        if (to - 1 > from) {
          betaGapStep(b, to - 1, prefetchEmis(to - 1));
        }
      }
      afterBeta(to - 1);
      if constexpr (SEQ) {
        for (int pos = to - 2; pos >= from; --pos) {
          betaSeqStep(b, pos);
          afterBeta(pos);
        }
      } else {
        EmisRegs ev;
        int rowNext = 0;
        if (to - 2 >= from) {
          ev = prefetchEmis(to - 1);
          rowNext = tStepRow[to - 1];
        }
        BetaHead head;
        bool primed = false;
        for (int pos = to - 2; pos >= from; --pos) {
          const int q = pos + 1;
          commitEmis(q, ev);
          const int row = rowNext;
          rowNext = -1;
          if (pos - 1 >= from) {
            ev = prefetchEmis(q - 1);
            rowNext = tStepRow[q - 1];
          }
          const int c = obsClass(q);
          beta_step<KT, KA>(K, b, w, tabs, row, &emisLds[q & 1][c * E4], cycW, head, primed, rowNext);
          primed = rowNext >= 0;
          afterBeta(pos);
        }
      }
    }

    FSMC_STAMP(cycB);
    // ------------------------------------------------------------------ pass A
    int cur = 4;      // open threshold level (0..3) or 4 = none
    int segStart = 0; // first site of the open segment
    float acc = 0.f;  // posteriorIBD
    float a[KA];

    auto emit = [&](const int s0, const int s1) {
      const unsigned idx = atomicAdd(&p.counters[1], 1u);
      float mean = 0.f, mapv = 0.f;
      if constexpr (TRACK) {
        segment_ages(K, p.ageThr, spsMem, tPi, tExpT, (p.flags & FSMC_WANT_MEAN) != 0,
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
    // LDS-DMA of one stored beta row (K4 x 1 KiB) into the landing zone: asynchronous, no VGPRs
    auto fetchBeta = [&](const float4* src) {
#pragma unroll
      for (int k4 = 0; k4 < K4; ++k4) {
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_global_load_lds(src + (size_t)k4 * kWave, &betaLds[k4 * kWave], 16, 0, 2 /* nt */);
#endif
      }
    };

    for (int j = 0; j < (nChunks > 0 ? nChunks : 0); ++j) {
      const int lo = from + j * C;
      const int hi = (lo + C < aEnd) ? lo + C : aEnd;
      if (!single) {
        // park the carried alpha while the chunk's betas are rebuilt
        if (j > 0) {
          store_vec<KT, KA>(K, saveA + lane, a);
        }
        {
          float b[KA];
          int pos;
          if (hi == to) {
            beta_init<KT, KA>(K, b);
            if constexpr (SEQ) {
              if (to - 1 > from) {
                betaGapStep(b, to - 1, prefetchEmis(to - 1));
              }
            }
            store_vec<KT, KA>(K, chunkbuf + slotOf(to - 1 - lo) * vecF4 + lane, b);
            pos = to - 2;
          } else {
            load_vec<KT, KA>(K, ckpt + (size_t)(j + 1) * vecF4 + lane, b);
            pos = hi - 1;
            if constexpr (SEQ) {
              commitEmis(hi, prefetchEmis(hi)); // the checkpoint is the stored vector of site hi: its rows next
            }
          }
          if constexpr (SEQ) {
            for (; pos >= lo; --pos) {
              betaSeqStep(b, pos);
              store_vec<KT, KA>(K, chunkbuf + (size_t)(pos - lo) * vecF4 + lane, b);
            }
          } else {
            EmisRegs ev;
            int rowNext = 0;
            if (pos >= lo) {
              ev = prefetchEmis(pos + 1);
              rowNext = tStepRow[pos + 1];
            }
            BetaHead head;
            bool primed = false;
            for (; pos >= lo; --pos) {
              const int q = pos + 1;
              commitEmis(q, ev);
              const int row = rowNext;
              rowNext = -1;
              if (pos - 1 >= lo) {
                ev = prefetchEmis(q - 1);
                rowNext = tStepRow[q - 1];
              }
              const int c = obsClass(q);
              beta_step<KT, KA>(K, b, w, tabs, row, &emisLds[q & 1][c * E4], cycW, head, primed, rowNext);
              primed = rowNext >= 0;
              const int rel = pos - lo;
              if (!HALF || (rel & 1) || pos == hi - 1) {
                store_vec<KT, KA>(K, chunkbuf + slotOf(rel) * vecF4 + lane, b);
              }
            }
          }
        }
        if (j > 0) {
          load_vec<KT, KA>(K, saveA + lane, a);
        }
      }

      FSMC_STAMP(cycR);
      // the wave's own stores of this chunk's betas must have landed before the DMA reads them back
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      FSMC_GCN_ASM("s_waitcnt vmcnt(0)" ::: "memory");
      fetchBeta(chunkbuf + lane);
      EmisRegs ev = prefetchEmis(lo);
      EmisRegs ev2 = ev; // HALF: the rows of the second site of a pair of sites
      if constexpr (SEQ) {
        commitEmis(lo, ev); // later sites are staged by the half-step of the site before them
      }
      if constexpr (HALF) {
        if (lo + 1 < hi) {
          ev2 = prefetchEmis(lo + 1);
        }
      }
      for (int pos = lo; pos < hi; ++pos) {
        // HALF: this site's beta row was not stored -- it is recomputed below from the row of site pos+1
        const bool rec = HALF && ((pos - lo) & 1) == 0 && pos + 1 < hi;

        if constexpr (SEQ) {
          if (pos < to - 1) {
            ev = prefetchEmis(pos + 1);
          }
        } else if constexpr (HALF) {
          // sites are taken two at a time: stage the rows of both (the beta step back from pos+1 needs them
          // before the alpha step into pos+1 does), and request the next two
          if (((pos - lo) & 1) == 0) {
            commitEmis(pos, ev);
            if (rec) {
              commitEmis(pos + 1, ev2);
            }
            if (pos + 2 < hi) {
              ev = prefetchEmis(pos + 2);
            }
            if (pos + 3 < hi) {
              ev2 = prefetchEmis(pos + 3);
            }
          }
        } else {
          commitEmis(pos, ev);
          if (pos + 1 < hi) {
            ev = prefetchEmis(pos + 1);
          }
        }
        const int c = obsClass(pos);
        const float4* e = &emisLds[pos & 1][c * E4];
        if (pos == from) {
          alpha_init<KT, KA>(K, a, tPi, e);
        } else {
          const int row = tStepRow[pos];
          alpha_step<KT, KA>(K, a, w, tabs, row, e, cycW);
        }
        if constexpr (SEQ) {
          // what the reference's alpha buffer holds for this site: alpha after the un-normalised half-step
          // across the gap to the next site (HMM.cpp:764-767); the last site of the window keeps its alpha
          if (pos < to - 1) {
            commitEmis(pos + 1, ev);
            const int row = tRowGapF[pos + 1];
            alpha_step<KT, KA, false>(K, a, w, tabs, row, &emisLds[(pos + 1) & 1][3 * E4],
                                      cycW);
          }
        }

        // combine with beta of this site (landed in LDS) and normalise (HMM.cpp:672-691)
        FSMC_GCN_ASM("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        float sumq = 0.f;
        if (rec) {
          // the landing zone holds beta of site pos+1: one beta step back gives this site's row (the same
          // operations, on the same bits, as the pass that stored its neighbours), combined from registers
          float b[KA];
#pragma unroll
          for (int k4 = 0; k4 < K4; ++k4) {
            const float4 o = betaLds[k4 * kWave + lane];
            b[4 * k4] = o.x;
            if (4 * k4 + 1 < K) b[4 * k4 + 1] = o.y;
            if (4 * k4 + 2 < K) b[4 * k4 + 2] = o.z;
            if (4 * k4 + 3 < K) b[4 * k4 + 3] = o.w;
          }
          const int q = pos + 1;
          const int cq1 = obsClass(q);
          const int rowq = tStepRow[q];
          BetaHead head;
          beta_step<KT, KA>(K, b, w, tabs, rowq, &emisLds[q & 1][cq1 * E4], cycW, head);
#pragma unroll
          for (int k = 0; k < K; k += 2) {
            if (KT > 0 && k + 1 < K) { // products two states at a time, the sum in state order
              const f32x2 av = {a[k], a[k + 1]};
              const f32x2 bv = {b[k], b[k + 1]};
              const f32x2 q = av * bv;
              w[k] = q.x;
              w[k + 1] = q.y;
              sumq = sumq + q.x;
              sumq = sumq + q.y;
            } else {
#pragma unroll
              for (int kk = k; kk < k + 2; ++kk) {
                if (kk < K) {
                  w[kk] = a[kk] * b[kk];
                  sumq = sumq + w[kk];
                }
              }
            }
          }
        } else {
          constexpr int kCB = 8; // states per block of the combine
          const int NB = (K + kCB - 1) / kCB;
          auto loadB = [&](const int blk, float4& b0, float4& b1) {
            b0 = betaLds[(2 * blk) * kWave + lane];
            b1 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (2 * blk + 1 < K4) {
              b1 = betaLds[(2 * blk + 1) * kWave + lane];
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
              if (KT > 0 && k + 1 < K) {
                const f32x2 av = {a[k], a[k + 1]};
                const f32x2 bv = {pick(c0, c1, i), pick(c0, c1, i + 1)};
                const f32x2 q = av * bv;
                w[k] = q.x;
                w[k + 1] = q.y;
                sumq = sumq + q.x;
                sumq = sumq + q.y;
              } else {
#pragma unroll
                for (int ii = i; ii < i + 2; ++ii) {
                  const int kk = blk * kCB + ii;
                  if (kk < K) {
                    w[kk] = a[kk] * pick(c0, c1, ii);
                    sumq = sumq + w[kk];
                  }
                }
              }
            }
            __builtin_amdgcn_sched_barrier(0);
            c0 = n0;
            c1 = n1;
          }
        }
        const float cq = 1.0f / sumq;
        // every read of the landing zone has returned: request the next site's beta row
        FSMC_GCN_ASM("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (MODE != kModeSums && !rec && pos + 1 < hi) {
          fetchBeta(chunkbuf + slotOf(pos + 1 - lo) * vecF4 + lane);
        }

        if (MODE == kModePerPair) {
          // HMM::writePerPairOutput (HMM.cpp:1378-1409): mean = sum_k post*E[t_k] (k ascending from 0.f),
          // MAP = first strictly larger posterior
          const cfloat_p tCoal = (cfloat_p)p.expCoal;
          float mean = 0.f;
          float best = 0.f;
          int arg = 0;
#pragma unroll
          for (int k = 0; k < K; ++k) {
            const float post = w[k] * cq;
            mean = mean + post * tCoal[k];
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

        if (MODE == kModeSums) {
          // HMM::augmentSumOverPairs (HMM.cpp:1052-1081): per site and state, the batch's posteriors are summed
          // over pairs in batch order (local fp32 sum from 0.f), then added to the accumulator.  The K x 64 tile
          // is transposed through LDS (row stride 65 floats: conflict-free both ways); lane j then owns state j.
          float* const tile = reinterpret_cast<float*>(betaLds);
          // a ring slot whose rows are no longer needed: this site's in sequence mode, the other one otherwise
          unsigned char* const cls = reinterpret_cast<unsigned char*>(&emisLds[(SEQ ? pos : pos + 1) & 1][0]);
#pragma unroll
          for (int k = 0; k < K; ++k) {
            tile[k * 65 + lane] = w[k] * cq;
          }
          if (p.flags & FSMC_WANT_MAJOR_MINOR_SUMS) {
            cls[lane] = (unsigned char)c;
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          for (int kk = lane; kk < K; kk += kWave) {
            float s = 0.f, s00 = 0.f, s01 = 0.f, s11 = 0.f;
            for (int v = 0; v < nPairsInGroup; ++v) {
              const float q = tile[kk * 65 + v];
              s = s + q;
              if (p.flags & FSMC_WANT_MAJOR_MINOR_SUMS) {
                const int cv = cls[v]; // 0 het -> 01, 1 hom major -> 00, 2 hom minor -> 11
                if (cv == 2) {
                  s11 = s11 + q;
                } else if (cv == 1) {
                  s00 = s00 + q;
                } else {
                  s01 = s01 + q;
                }
              }
            }
            float* acc = p.sums + (size_t)blockIdx.x * 4 * p.sumsPlane + (size_t)pos * K + kk;
            if (p.flags & FSMC_WANT_SUMS) acc[0] += s;
            if (p.flags & FSMC_WANT_MAJOR_MINOR_SUMS) {
              acc[p.sumsPlane] += s00;
              acc[2 * p.sumsPlane] += s01;
              acc[3 * p.sumsPlane] += s11;
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          FSMC_GCN_ASM("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_wave_barrier();
          if (pos + 1 < hi) {
            fetchBeta(chunkbuf + slotOf(pos + 1 - lo) * vecF4 + lane);
          }
        }

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
            // (the states beyond the scan's only feed the per-state sums of open segments: scaled there, by the
            //  lanes that are inside a segment)
            const unsigned nPost = p.stateThr;
            // (guards instead of early exits: a data-dependent trip count would turn the register
            //  arrays into dynamically indexed scratch memory)
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
            for (int k4 = 0; k4 < K4; ++k4) {
              if ((unsigned)(4 * k4) < p.stateThr) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                  if (4 * k4 + i < K && (unsigned)(4 * k4 + i) < p.stateThr) s = s + w[4 * k4 + i];
                }
              }
            }
            const int level = s >= p.thr[0] ? 0 : s >= p.thr[1] ? 1 : s >= p.thr[2] ? 2 : s >= p.thr[3] ? 3 : 4;
            // a change of level (or a drop below every threshold) closes the open segment at pos-1
            if (valid && cur != 4 && level != cur) {
              emit(segStart, pos - 1);
            }
            const bool opening = level != 4 && level != cur;
            if constexpr (TRACK) {
              // per-state posterior sums of the open segment (sum_posterior_per_state, HMM.cpp:1212-1229): kept
              // in the wave's workspace, touched only by the lanes that are inside a segment at this site
              if (level != 4) {
#pragma unroll
                for (int k4 = 0; k4 < K4; ++k4) {
                  if ((unsigned)(4 * k4) < p.ageThr) {
                    float4 sv = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (!opening) {
                      sv = spsMem[(size_t)k4 * kWave];
                    }
                    // blocks the scan already normalised are taken as they are (x * 1.0f is exact)
                    const float sc = ((unsigned)(4 * k4) < nPost) ? 1.0f : cq;
                    sv.x = sv.x + w[4 * k4] * sc;
                    if (4 * k4 + 1 < K) sv.y = sv.y + w[4 * k4 + 1] * sc;
                    if (4 * k4 + 2 < K) sv.z = sv.z + w[4 * k4 + 2] * sc;
                    if (4 * k4 + 3 < K) sv.w = sv.w + w[4 * k4 + 3] * sc;
                    spsMem[(size_t)k4 * kWave] = sv;
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
              if (valid && cur != 4) {
                emit(segStart, pos);
              }
            }
          }
        }
      }
      FSMC_STAMP(cycA);
    }
#if defined(FSMC_PHASE_STAMPS)
    if (lane == 0 && p.phaseCycles) {
      atomicAdd(&p.phaseCycles[0], (unsigned long long)cycB);
      atomicAdd(&p.phaseCycles[1], (unsigned long long)cycR);
      atomicAdd(&p.phaseCycles[2], (unsigned long long)cycA);
      atomicAdd(&p.phaseCycles[3], 1ull);
      atomicAdd(&p.phaseCycles[4], (unsigned long long)cycW);
    }
#endif
  }
}


// Sum of the per-wave accumulator planes in slot order (fixed order => reproducible fp32 result).
__global__ void reduce_planes_kernel(const float* __restrict__ planes, float* __restrict__ out, size_t n, int nSlots,
                                     size_t slotStride)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int sl = 0; sl < nSlots; ++sl) {
      s = s + planes[(size_t)sl * slotStride + i];
    }
    out[i] = s;
  }
}

} // namespace fsmc
