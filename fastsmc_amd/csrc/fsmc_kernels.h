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

#include <type_traits>

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
constexpr int kMaxStates = 256; // the widest model a lane-per-pair or two-workgroups-per-CU member decodes (fsmc_instances.h)
constexpr int kMaxStatesW2 = 1024; // ... and the wave-group kernel with one workgroup per CU (four waves of 80 states, six to eight of 64;
                                    // beyond 512 states eight waves of 80 / 96 / 128 without landing zones)
constexpr int kMaxStatesAny = 4096; // ... and the any-K kernel (fsmc_kernels_any.h: K-vectors in the workspace)

enum Mode : int { kModeIbd = 0, kModeDump = 1, kModePerPair = 2, kModeSums = 3 };

struct KParams {
  int K;       // states
  int KP;      // padded row stride of the tables (multiple of 4)
  int S;       // sites
  int W;       // 64-bit words per haplotype row
  int nGroups;
  int groupBase; // kModeSums: this launch decodes batches groupBase .. groupBase + gridDim.x - 1, one per wave;
                 // other modes: which of `counters` is this launch's queue head
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
  const float* ghostMask; // [KP] 1.0f for the K real states, 0.0f for the padding states (padded family members)
  const int* stepRow; // [S] row of the step into site q (array mode); sequence mode: the site step, forward
  const int* rowGapF; // sequence mode only: rows of the half-step across the gap (q-1, q), forward
  const int* rowSiteB; //                     site step, backward (out of site q)
  const int* rowGapB;  //                     gap half-step, backward
  const float4* emis3; // [S][3][KP/4]: emission rows for obs class het / hom-major / hom-minor
  const unsigned long long* haps; // [nHaps][W]
  const fsmc_pair* pairs;
  const fsmc_group* groups;
  unsigned* counters; // [groupBase] head of the group queue the launch pulls from (0, or 2 beside another launch), [1] IBD record count
  float4* ws;         // workspace, wsSlot float4 per resident wave
  size_t wsSlot;
  unsigned stateThr, ageThr;
  unsigned spsLds; // decode_kernel, TRACK: the launch carries dynamic LDS for the open segments' per-state sums
  float thr[4];       // {1000,100,10,1} * probabilityThreshold, evaluated in fp32 (HMM.cpp:1226...)
  fsmc_ibd_record* recs;
  unsigned recCap;
  float* dumpOut;             // kModeDump
  const size_t* dumpOffsets;  // [nGroups] float offsets
  float* ppMean;              // kModePerPair: [nPairs][S]
  int* ppMap;                 // kModePerPair: [nPairs][S]
  const float* expCoal;       // kModePerPair: [KP]
  unsigned long long* phaseCycles; // diagnostic builds (-DFSMC_PHASE_STAMPS): [0] pass B, [1] rebuild, [2] alpha sweep, [3] groups
  float* sums;                // kModeSums: one plane [S][K] (x4 with the 00/01/11 sums) per wave of the launch
  size_t sumsPlane;           // floats per plane
  size_t sumsSlot;            // floats per slot (wave / workgroup of the launch): one plane, or four with the 00/01/11 sums
  const unsigned* batchFirst; // kModeSums: [nBatches + 1] first group of every batch (null: every group is a batch)
  int residentChunks;         // chunked windows, array mode: the first this-many chunks of a window keep the rows pass B
                              // computes for them (a chunk buffer each) and skip the rebuild pass
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
#if defined(FSMC_DIAG_NOSMEM)
  FSMC_GCN_ASM("; no load %0 %1 %2" : "=s"(v) : "s"(p), "i"(byteOff));
#else
  FSMC_GCN_ASM("s_load_dwordx8 %0, %1, %2" : "=s"(v) : "s"(p), "i"(byteOff));
#endif
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
  static __device__ __forceinline__ T loadAt(cfloat_p p, const int firstState) { return sload4(p, firstState * 4); }
};
template <> struct SV<8> {
  typedef f32x8 T;
  static __device__ __forceinline__ T loadAt(cfloat_p p, const int firstState) { return sload8(p, firstState * 4); }
};
template <> struct SV<16> {
  typedef f32x16 T;
  static __device__ __forceinline__ T loadAt(cfloat_p p, const int firstState)
  {
    f32x16 v = {};
#if defined(FSMC_DIAG_NOSMEM) // timing experiment only (results are wrong): operands are whatever the registers hold
    FSMC_GCN_ASM("; no load %0 %1 %2" : "=s"(v) : "s"(p), "i"(firstState * 4));
#else
    FSMC_GCN_ASM("s_load_dwordx16 %0, %1, %2" : "=s"(v) : "s"(p), "i"(firstState * 4));
#endif
    return v;
  }
};
__device__ __forceinline__ void swait(f32x4& a, f32x4& b, f32x4& c, f32x4& d)
{
  FSMC_GCN_ASM(FSMC_SWAIT_INSN : "+s"(a), "+s"(b), "+s"(c), "+s"(d));
}

#if defined(FSMC_WAIT_STAMPS)
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
// Two-state products and sums: one packed instruction each, or (-DFSMC_UNPACK, an experiment switch) two plain ones.
// Same IEEE operations either way.
__device__ __forceinline__ f32x2 pmul(const f32x2 a, const f32x2 b)
{
#if defined(FSMC_UNPACK)
  const f32x2 r = {a.x * b.x, a.y * b.y};
  return r;
#else
  return a * b;
#endif
}
__device__ __forceinline__ f32x2 padd(const f32x2 a, const f32x2 b)
{
#if defined(FSMC_UNPACK)
  const f32x2 r = {a.x + b.x, a.y + b.y};
  return r;
#else
  return a + b;
#endif
}
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
#if defined(FSMC_DIAG_NOEMIS) // timing experiment only (results are wrong): no LDS read, the values are undefined
    FSMC_GCN_ASM("; no read" : "=v"(r.v[j].x), "=v"(r.v[j].y), "=v"(r.v[j].z), "=v"(r.v[j].w) : "v"(e));
#else
    r.v[j] = e[blk * (N / 4) + j];
#endif
  }
  return r;
}

// Diagnostic builds only (never in the shipped library): where a wave's cycles go.
//   -DFSMC_WAIT_STAMPS    cycles parked in the operand waits of the step functions
//   -DFSMC_REGION_STAMPS  cycles per code region: FSMC_END(dg, id) charges the time since the previous stamp to region
//                         id (s_memtime, low 32 bits; the accumulators are flushed once per group)
constexpr int kDiagRegions = 30; // (the wave-group kernel stamps 30 regions per wave role: fsmc_kernels_w2.h)
constexpr int kPhaseSlots = 8 + 4 * kDiagRegions; // entries of KParams::phaseCycles
struct Diag {
  long long waitCycles = 0;
#if defined(FSMC_REGION_STAMPS)
  unsigned acc[kDiagRegions] = {};
  unsigned last = 0;
#endif
};
#if defined(FSMC_REGION_STAMPS)
#define FSMC_END(dg, id)                                                                                               \
  do {                                                                                                                 \
    const unsigned now_ = (unsigned)__builtin_readcyclecounter();                                                      \
    (dg).acc[id] += now_ - (dg).last;                                                                                  \
    (dg).last = now_;                                                                                                  \
  } while (0)
#else
#define FSMC_END(dg, id) ((void)0)
#endif

// ---------------------------------------------------------------------------------------------
// The two steps, K a compile-time parameter (the kernel family, fsmc_instances.h).  One step of the backward recursion
// for one pair is HMM.cpp:957-1016 (NO_SSE association), of the forward recursion HMM.cpp:799-830 followed by the
// per-site scaling (HmmUtils.cpp:102-151).  gfx950 multiplies / adds two fp32 values per lane in one
// VALU instruction (v_pk_mul_f32, v_pk_add_f32) when both sit in an aligned register pair.  Every operation of a
// step that is not part of a first-order recurrence is done for states (k, k+1) at once; the recurrences (BU, BL,
// AU, the suffix sum, the scaling sum) stay scalar and sequential.  Each value is produced by the same IEEE
// operation on the same operands as in the scalar step (no FMA, no re-association), so the results are
// bit-identical; only ~7.5 instead of 11 VALU instructions are issued per state.
//
// Operand delivery.  One operand block = kKB (beta) / kKBF (alpha) states: the block's table values
// (scalar loads into SGPRs, inline asm) AND this lane's emission values of the same states (LDS reads) are
// requested together, a whole block ahead, right after the single wait that opens the block before.  That wait is the
// s_waitcnt BUILTIN, not inline asm: the compiler's wait-count pass sees it, marks its own LDS reads complete and
// inserts no counted lgkmcnt wait of its own inside a step.  (It cannot see the asm scalar loads, which share the
// counter and return out of order: a counted wait it placed for an LDS value right behind a freshly issued prefetch
// waited for that prefetch as well -- every block paid a full scalar-load latency.)
//
// Ghost states (GHOST = true: K < KT, KT a multiple of kKPad).  States K..KT-1 have zero table, emission and prior
// entries; their values stay exactly +0 through every operation and a sum that adds +0 in state order is the same
// sum.  The one exception, beta'[k] = (BL + D*vec) + BU = BL for a ghost, is multiplied by a 1/0 mask row (states of
// the last operand block only) before the scaling sum -- x * 1.0f is exact.  The reference's boundary cases
// (BU[K-1] = 0, no B term for the last state) come out of the same arithmetic: U*0 + RR*0 and B*0.
//
// Backward: the term U[k]*vec[k+1] of BU[k] is taken from T[m] = Ush[m]*vec[m] with Ush[m] = U[m-1]
// (a second copy of the U table shifted by one state), so that both factors share a state index.
__device__ __forceinline__ void waitLgkm0()
{
  __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0), vmcnt / expcnt left alone (gfx9 encoding)
}
__device__ __forceinline__ void waitVm0()
{
  __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
}
// After a wait: the registers of the scalar loads it covered are final.  Input-only on purpose -- an in/out ("+s")
// operand is a new value to the register allocator, which may then copy the load's destination into another
// register IN FRONT of the wait (observed: s_mov_b64 of sixteen in-flight SGPRs).  What keeps the uses behind the
// wait is the __builtin_amdgcn_sched_barrier(0) that follows every wait + prefetch group.
template <typename V> __device__ __forceinline__ void landed(const V& a, const V& b)
{
  FSMC_GCN_ASM("" ::"s"(a), "s"(b));
}
template <typename V> __device__ __forceinline__ void landed(const V& a, const V& b, const V& c, const V& d)
{
  FSMC_GCN_ASM("" ::"s"(a), "s"(b), "s"(c), "s"(d));
}
template <typename V> __device__ __forceinline__ void landed(const V& a)
{
  FSMC_GCN_ASM("" ::"s"(a));
}
// The wait that opens an operand block.  The scheduling barrier right behind it keeps register copies the allocator
// makes for the code below (post-RA scheduling included) from being hoisted in front of the wait.
#if defined(FSMC_WAIT_STAMPS) // (costs more than the waits it measures: its own switch, next to FSMC_PHASE_STAMPS)
#define FSMC_WAIT_OPERANDS(acc)                                                                                        \
  do {                                                                                                                 \
    const long long t0_ = (long long)clock64();                                                                        \
    waitLgkm0();                                                                                                       \
    (acc) += (long long)clock64() - t0_;                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
  } while (0)
#else
#define FSMC_WAIT_OPERANDS(acc)                                                                                        \
  do {                                                                                                                 \
    waitLgkm0();                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
  } while (0)
#endif

// Line warm-up generalised: COUNT (0..4) dwords, one per 64-byte line FIRST, FIRST+1, ... of the row at firstState.
template <int FIRST, int COUNT> __device__ __forceinline__ void touchRow(Touched& t, cfloat_p p, const int firstState)
{
  constexpr int STEP = 16;
  if constexpr (COUNT >= 4) {
    touchLines<FIRST, STEP>(t, p, firstState);
  } else if constexpr (COUNT == 3) {
    FSMC_GCN_ASM("s_load_dword %0, %3, %4\n\ts_load_dword %1, %3, %5\n\ts_load_dword %2, %3, %6"
                 : "=&s"(t.r[0]), "=&s"(t.r[1]), "=&s"(t.r[2])
                 : "s"(p), "i"((firstState + FIRST * STEP) * 4), "i"((firstState + (FIRST + 1) * STEP) * 4),
                   "i"((firstState + (FIRST + 2) * STEP) * 4));
  } else if constexpr (COUNT == 2) {
    FSMC_GCN_ASM("s_load_dword %0, %2, %3\n\ts_load_dword %1, %2, %4"
                 : "=&s"(t.r[0]), "=&s"(t.r[1])
                 : "s"(p), "i"((firstState + FIRST * STEP) * 4), "i"((firstState + (FIRST + 1) * STEP) * 4));
  } else if constexpr (COUNT == 1) {
    FSMC_GCN_ASM("s_load_dword %0, %1, %2" : "=&s"(t.r[0]) : "s"(p), "i"((firstState + FIRST * STEP) * 4));
  }
}
template <int COUNT> __device__ __forceinline__ void heldRow(const Touched& t)
{
  if constexpr (COUNT >= 4) {
    FSMC_GCN_ASM("" ::"s"(t.r[0]), "s"(t.r[1]), "s"(t.r[2]), "s"(t.r[3]));
  } else if constexpr (COUNT == 3) {
    FSMC_GCN_ASM("" ::"s"(t.r[0]), "s"(t.r[1]), "s"(t.r[2]));
  } else if constexpr (COUNT == 2) {
    FSMC_GCN_ASM("" ::"s"(t.r[0]), "s"(t.r[1]));
  } else if constexpr (COUNT == 1) {
    FSMC_GCN_ASM("" ::"s"(t.r[0]));
  }
}

// What a backward step has in flight when it opens: its first operand block (the top states of Ush and RR), the
// warm-up of those rows' other lines and the emission values of the same states.  Requested by beta_issue_pk ahead of
// operand-free work (the scaling multiply and the row store of the step before), consumed by beta_core_pk.
// Operand loads of the packed steps.  SY = false: asynchronous (the value is final behind the next operand wait; the
// compiled code is checked for instructions that touch the destination earlier, tools/check_inflight_sgprs.py).
// SY = true: load and wait in ONE asm statement -- nothing can come between, at the price of the latency at every
// block.  The sequence-mode kernels use it: their extra live scalars made the register allocator spill operand
// blocks right behind their loads (v_writelane of registers still in flight) in some instantiations.
template <int N, bool SY> struct LD {
  typedef typename SV<N>::T T;
  static __device__ __forceinline__ T loadAt(cfloat_p p, const int firstState)
  {
    if constexpr (!SY) {
      return SV<N>::loadAt(p, firstState);
    } else {
      T v = {};
      if constexpr (N == 16) {
        FSMC_GCN_ASM("s_load_dwordx16 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p), "i"(firstState * 4));
      } else if constexpr (N == 8) {
        FSMC_GCN_ASM("s_load_dwordx8 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p), "i"(firstState * 4));
      } else {
        FSMC_GCN_ASM("s_load_dwordx4 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p), "i"(firstState * 4));
      }
      return v;
    }
  }
};

template <int KT> struct BetaOps {
  typename SV<kKB>::T u, rr;
  Touched tu, trr;
  EmisBlk<kKB> em;
};

template <int KT, bool SY = false>
__device__ __forceinline__ void beta_issue_pk(BetaOps<KT>& o, cfloat_p rowSet, const float4* e)
{
  constexpr int KPc = ((KT + kKPad - 1) / kKPad) * kKPad;
  constexpr int NB = (KT + kKB - 1) / kKB;
  constexpr int NLINES = KPc / 16;
  o.u = LD<kKB, SY>::loadAt(rowSet, kRowUsh * KPc + (NB - 1) * kKB);
  o.rr = LD<kKB, SY>::loadAt(rowSet, kRowRR * KPc + (NB - 1) * kKB);
  if constexpr (kTouch && !SY) {
    touchRow<0, NLINES - 1>(o.tu, rowSet, kRowUsh * KPc);
    touchRow<0, NLINES - 1>(o.trr, rowSet, kRowRR * KPc);
  }
  o.em = readEmis<kKB>(e, NB - 1);
}

// The recurrences of one backward step: on entry b = beta of site pos+1 (scaled), on exit w = the un-normalised
// beta of site pos and the return value its sum over the states (k ascending from 0.f); b is used up.
template <int KT, int KA, bool GHOST, bool SY = false>
__device__ __forceinline__ float beta_core_pk(float (&b)[KA], float (&w)[KA], BetaOps<KT>& ops, cfloat_p rowSet,
                                              const float4* e, cfloat_p ghostMask, Diag& dg)
{
  constexpr int K = KT;
  // the five table rows of one key sit side by side (RowSet): one base register, block offsets as immediates
  constexpr int KPc = ((KT + kKPad - 1) / kKPad) * kKPad;
  typedef typename SV<kKB>::T SVec;
  constexpr int NB = (K + kKB - 1) / kKB; // operand blocks (scalar loads + emission values, one block ahead)
  constexpr int R = kKB / 8;              // sub-blocks of 8 states per operand block
  constexpr int NLINES = KPc / 16;
  static_assert(kKB == 16, "line warm-up and ghost masking are laid out for operand blocks of one 64-byte line");
  static_assert(!GHOST || K % kKB == 0, "ghost padding fills whole operand blocks");
  // (no initialisers: a copy of a register whose scalar load is still in flight would read garbage; every one of
  //  these is assigned behind a wait before it is read -- the block loops are fully unrolled)
  SVec u, rr, nu, nrr;
  EmisBlk<kKB> em, nem;
  SVec d, bt, nd, nbt, mk;
  Touched td, tbt;
  float tcarry = 0.f; // T of the first state of the sub-block above
#pragma unroll
  for (int blk = NB - 1; blk >= 0; --blk) {
    FSMC_WAIT_OPERANDS(dg.waitCycles);
    if (blk == NB - 1) {
      landed(ops.u, ops.rr);
      if constexpr (kTouch && !SY) {
        heldRow<NLINES - 1>(ops.tu);
        heldRow<NLINES - 1>(ops.trr);
      }
      u = ops.u;
      rr = ops.rr;
      em = ops.em;
    } else {
      landed(nu, nrr);
      u = nu;
      rr = nrr;
      em = nem;
    }
    if (blk > 0) {
      nu = LD<kKB, SY>::loadAt(rowSet, kRowUsh * KPc + (blk - 1) * kKB);
      nrr = LD<kKB, SY>::loadAt(rowSet, kRowRR * KPc + (blk - 1) * kKB);
      nem = readEmis<kKB>(e, blk - 1);
    } else {
      d = LD<kKB, SY>::loadAt(rowSet, kRowD * KPc);
      bt = LD<kKB, SY>::loadAt(rowSet, kRowB * KPc);
      if constexpr (kTouch && !SY) {
        touchRow<1, NLINES - 1>(td, rowSet, kRowD * KPc);
        touchRow<1, NLINES - 1>(tbt, rowSet, kRowB * KPc);
      }
      if constexpr (GHOST && NB == 1) {
        mk = LD<kKB, SY>::loadAt(ghostMask, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int sbi = R - 1; sbi >= 0; --sbi) {
      const int sb = blk * R + sbi;
      const int o = sbi * 8; // offset of this sub-block inside the operand block
      if (sb * 8 >= K) {
        continue;
      }
      float T[9];
      T[8] = tcarry;
      // vec[k] = beta[k]*e[k] (kept in b), T[k] = U[k-1]*vec[k]
#pragma unroll
      for (int i = 0; i < 8; i += 2) {
        const int k = sb * 8 + i;
        if (k + 1 < K) {
          f32x2 v = {b[k], b[k + 1]};
          v = pmul(v, em.pair(o + i));
          const f32x2 t = pmul(pairOf(u, o + i), v);
          b[k] = v.x;
          b[k + 1] = v.y;
          T[i] = t.x;
          T[i + 1] = t.y;
        } else if (k < K) {
          b[k] = b[k] * em.at(o + i);
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
    }
  }
  FSMC_END(dg, 2);
  // ascending: BL[k] = BL[k-1] + B[k-1]*vec[k-1];  beta'[k] = (BL[k] + D[k]*vec[k]) + BU[k]
  float BL = 0.f;
  float sum = 0.f;
#pragma unroll
  for (int blk = 0; blk < NB; ++blk) {
    FSMC_WAIT_OPERANDS(dg.waitCycles);
    if (blk > 0) {
      landed(nd, nbt);
      d = nd;
      bt = nbt;
    } else {
      landed(d, bt);
      if constexpr (kTouch && !SY) {
        heldRow<NLINES - 1>(td);
        heldRow<NLINES - 1>(tbt);
      }
    }
    if constexpr (GHOST) {
      if (blk == NB - 1) {
        landed(mk);
      }
    }
    if (blk + 1 < NB) {
      nd = LD<kKB, SY>::loadAt(rowSet, kRowD * KPc + (blk + 1) * kKB);
      nbt = LD<kKB, SY>::loadAt(rowSet, kRowB * KPc + (blk + 1) * kKB);
      if constexpr (GHOST) {
        if (blk + 1 == NB - 1) {
          mk = LD<kKB, SY>::loadAt(ghostMask, (NB - 1) * kKB);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < kKB; i += 2) {
      const int k = blk * kKB + i;
      if (k + 1 < K) {
        const f32x2 v = {b[k], b[k + 1]};
        const f32x2 dv = pmul(pairOf(d, i), v);
        const f32x2 bv = pmul(pairOf(bt, i), v);
        f32x2 bl;
        bl.x = BL;
        bl.y = BL + bv.x;
        f32x2 x = padd(bl, dv);
        const f32x2 bu = {w[k], w[k + 1]};
        x = padd(x, bu);
        if constexpr (GHOST) {
          if (blk == NB - 1) {
            x = pmul(x, pairOf(mk, i));
          }
        }
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
  FSMC_END(dg, 3);
  return sum;
}

// The operand-free tail of a step (HmmUtils.cpp:102-151: scal = 1.0f/sum, vec *= scal): v = w * (1.0f / sum).
// With sum = 1.0f it is an exact copy (a loaded checkpoint, the un-normalised half-steps of sequence mode).
template <int KT, int KA> __device__ __forceinline__ void scale_pk(float (&v)[KA], const float (&w)[KA], const float sum)
{
  constexpr int K = KT;
  const float c = 1.0f / sum;
  const f32x2 cc = {c, c};
#pragma unroll
  for (int k = 0; k < K; k += 2) {
    if (k + 1 < K) {
      const f32x2 x = {w[k], w[k + 1]};
      const f32x2 y = pmul(x, cc);
      v[k] = y.x;
      v[k + 1] = y.y;
    } else {
      v[k] = w[k] * c;
    }
  }
}

// A whole backward step in one piece (the sequence-mode passes and the recomputed rows of beta stride 2).
template <int KT, int KA, bool SCALE, bool GHOST, bool SY = false>
__device__ __forceinline__ void beta_step_pk(float (&b)[KA], float (&w)[KA], cfloat_p rowSet, const float4* e,
                                             cfloat_p ghostMask, Diag& dg)
{
  BetaOps<KT> ops;
  beta_issue_pk<KT, SY>(ops, rowSet, e);
  const float sum = beta_core_pk<KT, KA, GHOST, SY>(b, w, ops, rowSet, e, ghostMask, dg);
  if constexpr (SCALE) {
    scale_pk<KT, KA>(b, w, sum);
  } else {
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      b[k] = w[k];
    }
  }
}

// Forward.  The suffix sums are kept one slot down (w[k] = alphaC[k+1]) so that B[k]*alphaC[k+1] pairs up with the
// other products of state k; alphaC[0] is never used (HMM.cpp:799-830).
template <int KT, int KA, bool SCALE = true, bool SY = false>
__device__ __forceinline__ void alpha_step_pk(float (&a)[KA], float (&w)[KA], cfloat_p rowSet, cfloat_p cR,
                                              const float4* e, Diag& dg)
{
  constexpr int K = KT;
  constexpr int KPc = ((KT + kKPad - 1) / kKPad) * kKPad;
  static_assert(K >= 2, "packed step needs at least two states");
  typedef typename SV<kKBF>::T SVec;
  constexpr int NB = (K + kKBF - 1) / kKBF; // operand blocks (scalar loads + emission values, one block ahead)
  constexpr int NLINES = KPc / 16;
  SVec d = LD<kKBF, SY>::loadAt(rowSet, kRowD * KPc), bt = LD<kKBF, SY>::loadAt(rowSet, kRowB * KPc),
       u = LD<kKBF, SY>::loadAt(rowSet, kRowU * KPc), c4 = LD<kKBF, SY>::loadAt(cR, 0);
  Touched td, tbt, tu;
  if constexpr (kTouch && !SY) {
    touchRow<1, NLINES - 1>(td, rowSet, kRowD * KPc);
    touchRow<1, NLINES - 1>(tbt, rowSet, kRowB * KPc);
    touchRow<1, NLINES - 1>(tu, rowSet, kRowU * KPc);
  }
  SVec nd, nbt, nu, nc; // (assigned behind a wait before they are read; see beta_core_pk)
  EmisBlk<kKBF> em = readEmis<kKBF>(e, 0), nem;
  __builtin_amdgcn_sched_barrier(0);
  // operand-free: alphaC[k+1] = sum_{i>k} alpha[i], accumulated from the top (HMM.cpp:799-814)
  w[K - 2] = a[K - 1];
#pragma unroll
  for (int k = K - 2; k >= 1; --k) {
    w[k - 1] = w[k] + a[k];
  }
  float AU = 0.f;
  float sum = 0.f;
  FSMC_END(dg, 5);
#pragma unroll
  for (int blk = 0; blk < NB; ++blk) {
    FSMC_WAIT_OPERANDS(dg.waitCycles);
    if (blk > 0) {
      landed(nd, nbt, nu, nc);
      d = nd;
      bt = nbt;
      u = nu;
      c4 = nc;
      em = nem;
    } else {
      landed(d, bt, u, c4);
      if constexpr (kTouch && !SY) {
        heldRow<NLINES - 1>(td);
        heldRow<NLINES - 1>(tbt);
        heldRow<NLINES - 1>(tu);
      }
    }
    if (blk + 1 < NB) {
      nd = LD<kKBF, SY>::loadAt(rowSet, kRowD * KPc + (blk + 1) * kKBF);
      nbt = LD<kKBF, SY>::loadAt(rowSet, kRowB * KPc + (blk + 1) * kKBF);
      nu = LD<kKBF, SY>::loadAt(rowSet, kRowU * KPc + (blk + 1) * kKBF);
      nc = LD<kKBF, SY>::loadAt(cR, (blk + 1) * kKBF);
      nem = readEmis<kKBF>(e, blk + 1);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < kKBF; i += 2) {
      const int k = blk * kKBF + i;
      if (k + 1 < K - 1) {
        const f32x2 av = {a[k], a[k + 1]};
        const f32x2 da = pmul(pairOf(d, i), av);
        const f32x2 ua = pmul(pairOf(u, i), av);
        const f32x2 ac = {w[k], w[k + 1]};
        const f32x2 bw = pmul(pairOf(bt, i), ac);
        f32x2 au;
        au.x = AU;
        au.y = ua.x + c4[i] * AU; // AU of state k+1
        f32x2 term = padd(au, da);
        term = padd(term, bw);
        const f32x2 ov = pmul(em.pair(i), term);
        w[k] = ov.x;
        w[k + 1] = ov.y;
        sum = sum + ov.x;
        sum = sum + ov.y;
        AU = ua.y + c4[i + 1] * au.y; // AU of state k+2
      } else {
#pragma unroll
        for (int ii = i; ii < i + 2; ++ii) {
          const int kk = blk * kKBF + ii;
          if (kk < K) {
            float term = AU + d[ii] * a[kk];
            if (kk < K - 1) {
              term = term + bt[ii] * w[kk];
            }
            w[kk] = em.at(ii) * term;
            sum = sum + w[kk];
            if (kk < K - 1) {
              AU = u[ii] * a[kk] + c4[ii] * AU;
            }
          }
        }
      }
    }
  }
  FSMC_END(dg, 6);
  if constexpr (SCALE) {
    scale_pk<KT, KA>(a, w, sum);
  } else {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      a[k] = w[k];
    }
  }
  FSMC_END(dg, 7);
}



// The tables as the kernel addresses them: `row` selects the key of the RowSet copy.
struct Tables {
  cfloat_p rowSets, cR, ghostMask;
};

// compile-time K that is a whole number of operand blocks = a padded family member (K <= KT real states)
template <int KT> constexpr bool kGhost = KT % kKPad == 0;

template <int KT> __device__ __forceinline__ cfloat_p rowSetOf(const Tables& t, const int row)
{
  constexpr int KPc = ((KT + kKPad - 1) / kKPad) * kKPad;
  return t.rowSets + (size_t)row * (kRowSetParts * KPc);
}

// Sequence mode loaded its operands synchronously in rounds 2-3 (see LD: its extra live scalars made the allocator
// spill operand blocks right behind their loads in SOME instantiations).  The in-flight check of every compiled
// member (tools/check_inflight_sgprs.py, tests/test_isa_hazards.py) shows which: the 16- and 32-state members; all the
// others -- among them the 69-state member of the reference's own files -- are clean with the asynchronous loads of
// array mode, one operand block ahead, and use them -- in the IBD decode, the product's path; the three other consumers
// of sequence mode (posterior dump, per-pair rows, sums over pairs) keep the synchronous loads: which of their
// instantiations are clean moved with an unrelated change to the sums consumer (64-state member), and they are not
// worth a list of their own.  (-DFSMC_SEQ_SYNC_LOADS: synchronous everywhere, as before.)
// The LDS home of the segments' per-state sums (KParams::spsLds) is built into the array-mode IBD decode with one group
// per wave -- the kernel of launches that can be smaller than the chip.
template <int MODE, bool SEQ, bool DUAL> constexpr bool kSpsLdsBuilt = MODE == kModeIbd && !SEQ && !DUAL;
#if defined(FSMC_SEQ_SYNC_LOADS)
template <bool SEQ, int KT, int MODE> constexpr bool kSeqSyncLoads = SEQ;
#else
template <bool SEQ, int KT, int MODE> constexpr bool kSeqSyncLoads = SEQ && (KT <= 32 || MODE != kModeIbd);
#endif
template <int KT, int KA, bool SCALE = true, bool SY = false>
__device__ __forceinline__ void beta_step(const int K, float (&b)[KA], float (&w)[KA], const Tables& t, const int row,
                                          const float4* e, Diag& dg)
{
  (void)K;
  beta_step_pk<KT, KA, SCALE, kGhost<KT>, SY>(b, w, rowSetOf<KT>(t, row), e, t.ghostMask, dg);
}

template <int KT, int KA, bool SCALE = true, bool SY = false>
__device__ __forceinline__ void alpha_step(const int K, float (&a)[KA], float (&w)[KA], const Tables& t, const int row,
                                           const float4* e, Diag& dg)
{
  (void)K;
  alpha_step_pk<KT, KA, SCALE, SY>(a, w, rowSetOf<KT>(t, row), t.cR, e, dg);
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
// (Kreal < K for a padded family member: the ghost states start at +0 and add +0 to the sum)
template <int KT, int KA> __device__ __forceinline__ void beta_init(const int K, const int Kreal, float (&b)[KA])
{
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    b[k] = (!kGhost<KT> || k < Kreal) ? 1.0f : 0.f;
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
// Addressing: `row` is the WAVE-UNIFORM address of the vector (lane 0, states 0..3) and laneOff = 16 * lane.  Each
// access is then "scalar base + 32-bit lane offset + immediate" (the saddr form of the global instructions): no
// per-row 64-bit address vectors.  (Kept as per-lane pointers the compiler hoisted eighteen of them out of the site
// loops, spilled them, and reloaded each in front of its store behind an s_waitcnt vmcnt(0) -- every row went out
// as fifteen serialised memory round trips, which was 50 % of the kernel's time.)
// (the result is typed as a GLOBAL pointer: rebuilt from integers the address space cannot be inferred, and a
//  generic pointer makes these flat_* instructions, which count in lgkmcnt as well and complete out of order)
typedef char __attribute__((address_space(1))) * gchar_p;
typedef f32x4 __attribute__((address_space(1))) * gf32x4_p;
__device__ __forceinline__ gchar_p uniformPtr(const void* p)
{
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (gchar_p)(((unsigned long long)hi << 32) | lo);
}
// address of float4 group k4 of this lane inside the 1-KiB-per-group row at `base`
__device__ __forceinline__ gf32x4_p rowSlot(const gchar_p base, const int k4, const unsigned laneOff)
{
  return (gf32x4_p)(base + (size_t)k4 * (kWave * sizeof(float4)) + laneOff);
}
// LDS-DMA of 16 bytes per lane (lane i lands at lds + 16*i) as inline asm, for the two-slot rings.  Through the
// builtin the compiler knows that LDS is being written and -- the two slots of a ring being one array -- guards the
// reads of the OTHER slot with a vmcnt(0), i.e. waits for the request that was just issued for the next site.  As
// asm the request is opaque (a "memory" clobber keeps LDS accesses from moving across it) and is covered by the
// explicit vmcnt waits of the callers.  M0 carries the LDS address; the compiler re-initialises M0 before every use
// of its own, and the clobber tells it that this statement changes it.
#if defined(__clang__)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
#endif
__device__ __forceinline__ void dmaToLds(const gf32x4_p src, const void* ldsDst)
{
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned lds = (unsigned)(unsigned long long)(const char __attribute__((address_space(3)))*)ldsDst;
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds) : "memory", "m0");
#endif
}
#if defined(__clang__)
#pragma clang diagnostic pop
#endif

template <int KT, int KA>
__device__ __forceinline__ void store_vec(const int K, float4* row, const unsigned laneOff, const float (&v)[KA])
{
  const gchar_p base = uniformPtr(row);
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
    __builtin_nontemporal_store(ov, rowSlot(base, k4, laneOff));
  }
}

template <int KT, int KA>
__device__ __forceinline__ void load_vec(const int K, const float4* row, const unsigned laneOff, float (&v)[KA])
{
  const gchar_p base = uniformPtr(row);
  const int K4 = (K + 3) >> 2;
#pragma unroll
  for (int k4 = 0; k4 < K4; ++k4) {
    const f32x4 ov = __builtin_nontemporal_load(rowSlot(base, k4, laneOff));
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

// The IBD scan's sum over the states below the threshold, k ascending from 0.f (HMM.cpp:1207-1224), over a K-vector in
// registers; the states it covers are scaled in place on the way.  The walk leaves at the first block of four states
// beyond the threshold -- one taken branch per site -- and inside the block that holds the threshold the states beyond it
// add +0.f, which leaves the non-negative sum unchanged.  `nPost` must be a value the compiler cannot prove
// loop-invariant (launderScalar): the compare of every state was otherwise hoisted out of the site loop as a lane mask in
// an SGPR pair, all of them spilled -- two v_readlane per state and site in front of every v_cndmask.
__device__ __forceinline__ unsigned launderScalar(unsigned v)
{
  FSMC_GCN_ASM("" : "+s"(v));
  return v;
}
template <int KA, int K, int K4>
__device__ __forceinline__ void scanBlocks(float (&w)[KA], float& s, const float cq, const unsigned nPost)
{
#pragma unroll
  for (int k4 = 0; k4 < K4; ++k4) {
    if ((unsigned)(4 * k4) >= nPost) {
      break;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (4 * k4 + i < K) {
        w[4 * k4 + i] = w[4 * k4 + i] * cq;
        s = s + ((unsigned)(4 * k4 + i) < nPost ? w[4 * k4 + i] : 0.f);
      }
    }
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
// Waves per SIMD the register allocation leaves room for: two 256-register waves up to 80 states; the members for 96,
// 112 and 128 states keep their two or three K-vectors in registers only with the whole 512-entry file (one wave).
constexpr int minWavesPerSimd(const int KT)
{
  return KT >= 70 ? 1 : 2; // (the exact members between 69 and 80 states: three K-vectors no longer fit 256 registers)
}
//
// DUAL: two half-groups per wavefront (hashing mode: a batch is 32 pairs, half a wave).  Lanes 0..31 decode half A, lanes
// 32..63 half B, each over ITS OWN decode and scan window; the wave walks the union of the two windows and a lane is
// (re)initialised where its own window opens -- beta at site to-1 of its window, alpha at site from -- so within its
// window every value is produced by exactly the operations of a stand-alone decode (what a lane computes outside its
// window is overwritten before it is used and never reaches an output).  Single-chunk layout, beta stride 1 or 2 (with
// stride 2 a window may end at a site whose row is recomputed in the alpha sweep: the re-initialisation is repeated
// there); the work list is an array of item = {group A, group B} (B may be empty) prepared by the host library.
template <int KT, int MODE, bool TRACK, bool SEQ, bool HALF, bool DUAL = false>
__global__ __launch_bounds__(kWave, minWavesPerSimd(KT)) void decode_kernel(const KParams p)
{
  static_assert(!HALF || (!SEQ && (MODE == kModeIbd || MODE == kModeSums)),
                "beta stride 2 is built for the array-mode IBD decode and the array-mode sums over pairs");
  static_assert(KT > 0 && KT <= kMaxStates, "a member of the kernel family (fsmc_instances.h)");
  static_assert(!DUAL || (MODE == kModeIbd && !SEQ), "two half-groups per wave: array-mode IBD");
  // array mode: the backward loops are rotated (operand-free step tails overlap the next step's first operand requests)
  constexpr bool kRotate = !SEQ;
  constexpr int KA = KT;
  constexpr int K4A = (KA + 3) / 4;
  constexpr int E4A = ((KA + kKPad - 1) / kKPad) * (kKPad / 4); // float4 per emission row (rows padded to kKPad)
  constexpr int NC = SEQ ? 4 : 3;                         // emission rows per site: 3 observation classes (+ gap)
  constexpr int NL = (NC * E4A + kWave - 1) / kWave;      // float4 per lane to stage one site's rows
  constexpr int K = KT;
  // states of the model: < K for a padded family member (the states Kreal..K-1 are ghosts), otherwise K
  const int Kreal = kGhost<KT> ? p.K : K;
  constexpr int K4 = (K + 3) >> 2;
  const int KP = p.KP; // (a compile-time KP for fixed K measured 6 % slower on C2: keep the runtime value)
  const int E4 = KP >> 2;

  __shared__ float4 emisLds[2][NC * E4A]; // ring of two sites x three observation classes (+ the gap row)
  // landing zone of the next site's beta row (LDS-DMA); the sums consumer also transposes the K x 64 posterior tile
  // through it with a row stride of 65 floats
  constexpr int kLandF4 = (MODE == kModeSums && (KA * 65 + 3) / 4 > K4A * kWave) ? (KA * 65 + 3) / 4 : K4A * kWave;
  __shared__ float4 betaLds[kLandF4];
  // kModePerPair: the expected coalescence times (read in the consumer's loop over the states; as scalar loads they
  // were K values live at once -- fsmc_kernels_bidir.h, same place)
  __shared__ float coalLds[MODE == kModePerPair ? KA : 1];

  const int lane = threadIdx.x;
  const cfloat_p tPi = (cfloat_p)p.pi, tCR = (cfloat_p)p.cR, tExpT = (cfloat_p)p.expT;
  const Tables tabs = {(cfloat_p)p.rowSets, tCR, (cfloat_p)p.ghostMask};
  const cint_p tStepRow = (cint_p)p.stepRow;
  const cint_p tRowGapF = (cint_p)p.rowGapF, tRowSiteB = (cint_p)(SEQ ? p.rowSiteB : p.stepRow),
               tRowGapB = (cint_p)p.rowGapB;
  const size_t vecF4 = (size_t)K4 * kWave; // float4 per stored K-vector of a wave
  float4* const chunkbuf = p.ws + (size_t)blockIdx.x * p.wsSlot;
  float4* const ckpt = chunkbuf + (size_t)p.chunkRows * vecF4;
  float4* const saveA = ckpt + (size_t)(p.maxChunks + 2) * vecF4;
  float4* const saveS = saveA + vecF4;
  // Resident chunks (multi-chunk windows, array mode): pass B walks down to the window's first site, so the rows of the
  // window's FIRST chunks are the last it computes -- where the workspace has room (288 GB of HBM: the host plans up to
  // p.residentChunks extra chunk buffers per wave) pass B keeps them and pass A sweeps those chunks without rebuilding
  // them: the same rows, bit for bit (the rebuild repeats pass B's operations on the same operands).
  float4* const resBase = saveS + vecF4;
  // (array-mode IBD decode and sums over pairs, one group per wave: the paired kernel is chunked too, but keeps no
  //  resident chunks)
  constexpr bool kResidentBuilt = !SEQ && !DUAL && (MODE == kModeIbd || MODE == kModeSums);
  const int nResident = kResidentBuilt ? p.residentChunks : 0;
  const int C = p.chunk;
  // per-state posterior sums of the open segments (TRACK), one column per lane.  They live in the wave's workspace --
  // or, when the launch leaves LDS for them (fewer waves a CU than LDS could hold: the host passes K4 x 1 KiB of dynamic
  // LDS and sets p.spsLds), in LDS: a lane inside a segment reads and rewrites them at every site, a round trip to L2 per
  // sixteen states that a wave alone on its SIMD waits out in full (C1 shape: 27 % of the kernel).
  extern __shared__ float4 spsDyn[]; // [K4][64 lanes]
  const bool spsInLds = TRACK && kSpsLdsBuilt<MODE, SEQ, DUAL> && p.spsLds != 0;
  const float4* const spsMem = spsInLds ? (const float4*)(spsDyn + threadIdx.x) : (const float4*)(saveS + threadIdx.x);
  const unsigned laneOff = threadIdx.x * (unsigned)sizeof(float4); // this lane's byte offset inside a 1-KiB row
  // chunk-buffer slot of the row stored for the site at offset rel of its chunk
#if defined(FSMC_DIAG_SAMEROW)
  // timing experiment only (results are wrong): every stored row lands on the chunk buffer's first slot, so the
  // instruction stream is unchanged but the beta stream stays in the caches instead of going through HBM
  auto slotOf = [](const int rel) -> size_t { return (size_t)(rel & 0); };
#else
  auto slotOf = [](const int rel) -> size_t { return (size_t)(HALF ? (rel >> 1) : rel); };
#endif

  struct EmisRegs {
    float4 v[NL];
  };
  if (MODE == kModePerPair) {
    for (int k = lane; k < K; k += kWave) {
      coalLds[k] = (k < Kreal) ? p.expCoal[k] : 0.f;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }

  for (unsigned round = 0;; ++round) {
#if defined(FSMC_DIAG_GROUP_TIMES)
    const unsigned long long diagT0 = __builtin_amdgcn_s_memrealtime();
#endif
    unsigned g = 0;
    if (MODE == kModeSums) {
      // one BATCH per wave and launch: wave i writes the sums of batch groupBase + i into plane i, and the host adds the
      // planes to the accumulator one after the other -- the reference's order, batch by batch (HMM.cpp:1054-1073).  A
      // batch of more than 64 pairs is several consecutive groups (p.batchFirst: first group of every batch): the wave
      // decodes them in turn and every group continues the batch's running sums where the group before left them.
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
      if (lane == 0) {
        g = atomicAdd(&p.counters[p.groupBase], 1u);
      }
      g = __builtin_amdgcn_readfirstlane(g);
    }
    if (g >= (unsigned)p.nGroups) {
      break;
    }
    const cuint_p gw = (cuint_p)(p.groups + (DUAL ? 2 * (size_t)g : (size_t)g));
    const unsigned firstPair = gw[0];
    const int nPairsInGroup = (int)gw[1];
    int from = (int)gw[2];
    int to = (int)gw[3];
    int scanFrom = (int)gw[4];
    int aEnd = (MODE == kModeIbd) ? (int)gw[5] : to; // the alpha sweep stops here
    bool valid = lane < nPairsInGroup;
    unsigned pairIdx = firstPair + (valid ? (unsigned)lane : 0u);
    // DUAL: this lane's own windows, and the wave-uniform windows of the two halves
    int myFrom = from, myTo = to, mySF = scanFrom, myST = aEnd;
    int toA = to, toB = to, fromA = from, fromB = from, stA = aEnd, stB = aEnd;
    if constexpr (DUAL) {
      const int nB = (int)gw[7];
      if (nB > 0) {
        fromB = (int)gw[8];
        toB = (int)gw[9];
        stB = (int)gw[11];
        const bool half = lane >= 32;
        valid = half ? (lane - 32 < nB) : (lane < nPairsInGroup);
        pairIdx = half ? gw[6] + (valid ? (unsigned)(lane - 32) : 0u) : firstPair + (valid ? (unsigned)lane : 0u);
        myFrom = half ? fromB : fromA;
        myTo = half ? toB : toA;
        mySF = half ? (int)gw[10] : scanFrom;
        myST = half ? stB : stA;
        from = fromA < fromB ? fromA : fromB;
        to = toA > toB ? toA : toB;
        scanFrom = scanFrom < (int)gw[10] ? scanFrom : (int)gw[10];
        aEnd = stA > stB ? stA : stB;
      } else {
        valid = lane < nPairsInGroup && lane < 32;
      }
    }
    const fsmc_pair pr = p.pairs[pairIdx];
    const unsigned long long* rowA = p.haps + (size_t)pr.hap_a * p.W;
    const unsigned long long* rowB = p.haps + (size_t)pr.hap_b * p.W;

    const int nA = aEnd - from;
    const int nChunks = (nA + C - 1) / C;
    const bool single = nChunks <= 1;
    // Wave priority = how much of its group a wave still has in front of it.  Left alone, the two waves of a SIMD do not
    // share it evenly: the older one is served first and runs its groups in half the time of its partner (C3 windows:
    // 0.7 s against 1.4 s a group, per-group stamps of a diagnostic build) -- the same throughput while the queue is full,
    // but when it runs dry the favoured wave is done early and its partner finishes alone, on a SIMD that a lone wave
    // does not fill.  With the wave that has MORE left in front the two finish together: 4096 groups of the C3 shape (what
    // each of four GPUs gets) 2291 -> 1927 ms, 2048 groups 1023 -> 980, C2 1827 -> 1810 (same box).
    __builtin_amdgcn_s_setprio(3); // a fresh group: everything remains (pass B is about a third of it)

    int wordIdx = -1;
    unsigned long long xw = 0, aw = 0;
    // observation class of this lane's pair at site q: 0 het, 1 hom major, 2 hom minor
    // (obsIsZero / obsIsTwo of HMM.cpp:647-652 folded into a row select)
    auto obsClass = [&](const int q) -> int {
      const int wi = q >> 6;
      if (__builtin_expect(wi != wordIdx, 0)) { // once per 64 sites
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

    // Array mode: a site's emission rows go straight from global memory into ring slot (q & 1) by LDS-DMA -- no
    // staging registers, no ds_write (kept in registers from one iteration to the next they were spilled: a wait
    // for the load's round trip plus a scratch store at every site).  Asynchronous: counts in vmcnt and is visible
    // to this wave's LDS reads behind a vmcnt wait that covers it.  The slot's previous rows must no longer be read.
    auto stageEmis = [&](const int q) {
      const gchar_p src = uniformPtr(p.emis3 + (size_t)q * (NC * E4));
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const int idx = lane + i * kWave;
        if (idx < NC * E4) {
          dmaToLds((gf32x4_p)(src + (size_t)i * (kWave * sizeof(float4)) + laneOff), &emisLds[q & 1][i * kWave]);
        }
      }
    };
    // Table row of the step into `site`: the indices of 64 consecutive sites sit in one VGPR (lane = site % 64,
    // one coalesced load per 64 sites) and are picked with v_readlane.  (As scalar loads, one per site, each was
    // waited for on the spot -- the register allocator had no SGPR to keep it in flight -- a round trip to L2 per
    // site in the loop heads.)
    int rowBlk = -1;
    int rowVec = 0;
    auto stepRowOf = [&](const int site) -> int {
      const int blk = site >> 6;
      if (__builtin_expect(blk != rowBlk, 0)) { // once per 64 sites
        const int idx = blk * kWave + lane;
        rowVec = p.stepRow[idx < p.S ? idx : p.S - 1];
        rowBlk = blk;
        // waited for here, once per 64 sites: otherwise the compiler, which cannot tell at the v_readlane below
        // whether this load is the pending one, waits for vmcnt(0) at EVERY site -- right behind the emission-row
        // request that was just issued
        waitVm0();
      }
      return __builtin_amdgcn_readlane(rowVec, site & (kWave - 1));
    };
    // Wait for the emission rows requested one iteration ago (prefetchEmis) while the beta-row stores issued behind
    // them may still be in flight: vector memory operations retire in order, so "at most K4 outstanding" means the
    // older request is done.  Explicit, because the compiler cannot count a conditional store burst and would wait
    // for vmcnt(0) -- i.e. for the stores to reach HBM -- at every site.
    auto waitEmisRows = [&](const bool storesBehind) {
      constexpr unsigned n = (unsigned)K4A; // (only the compile-time-K loops call this)
      if (storesBehind && n < 64) {
        __builtin_amdgcn_s_waitcnt(0x0F70 | (n & 15u) | ((n >> 4) << 14));
      } else {
        waitVm0();
      }
    };

    float w[KA];
    Diag cycW; // diagnostic builds only: cycles parked in operand waits, cycles per code region
#if defined(FSMC_REGION_STAMPS)
    cycW.last = (unsigned)__builtin_readcyclecounter();
#endif
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
      beta_step<KT, KA, false, kSeqSyncLoads<SEQ, KT, MODE>>(K, b, w, tabs, row, &emisLds[q & 1][3 * E4], cycW);
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
      beta_step<KT, KA, true, kSeqSyncLoads<SEQ, KT, MODE>>(K, b, w, tabs, row, &emisLds[q & 1][c * E4], cycW);
      if (gap) {
        betaGapStep(b, pos, ev);
      }
    };

    // ------------------------------------------------------------------ pass B
    {
      float b[KA];
      if constexpr (!kRotate) {
        beta_init<KT, KA>(K, p.K, b);
      }
      // Checkpoints (multi-chunk windows): beta at the first site of every chunk but the first -- slot j for site
      // from + j*C -- and at aEnd (slot nChunks) when the alpha sweep stops short of the window.  Pass B walks down,
      // so the next checkpoint is a countdown (no integer division per site).
      int ckJ = (aEnd < to) ? nChunks : nChunks - 1;
      int ckPos = (aEnd < to) ? aEnd : from + ckJ * C;
      auto afterBeta = [&](const int pos) -> bool { // true if a row went out
        if (single) {
          const int rel = pos - from;
          if (pos < aEnd && (!HALF || (rel & 1) || pos == aEnd - 1)) {
            store_vec<KT, KA>(K, chunkbuf + slotOf(rel) * vecF4, laneOff, b);
            return true;
          }
        } else {
          bool out = false;
          // (pos lies in chunk ckJ: the checkpoint countdown below IS the chunk pass B is in)
          if (kResidentBuilt && ckJ < nResident && pos < aEnd) { // a resident chunk: the row stays, at its chunk-local slot
            const int relc = pos - ckPos;
            const int hiJ = ckPos + C < aEnd ? ckPos + C : aEnd;
            if (!HALF || (relc & 1) || pos == hiJ - 1) {
              store_vec<KT, KA>(K, resBase + ((size_t)ckJ * p.chunkRows + slotOf(relc)) * vecF4, laneOff, b);
              out = true;
            }
          }
          if (__builtin_expect(pos == ckPos && ckJ >= 1, 0)) { // once per chunk
            store_vec<KT, KA>(K, ckpt + (size_t)ckJ * vecF4, laneOff, b);
            ckJ -= 1;
            ckPos = from + ckJ * C;
            out = true;
          }
          return out;
        }
        return false;
      };
      if constexpr (SEQ) {
        if (to - 1 > from) {
          betaGapStep(b, to - 1, prefetchEmis(to - 1));
        }
      }
      if constexpr (!kRotate) {
        afterBeta(to - 1);
      }
      if constexpr (SEQ) {
        for (int pos = to - 2; pos >= from; --pos) {
          betaSeqStep(b, pos);
          afterBeta(pos);
        }
      } else {
        if (to - 2 >= from) {
          stageEmis(to - 1);
        }
        if constexpr (kRotate) {
          // Rotated loop: the operand-free tail of a step (1/sum, the scaling multiply, the row store) runs while
          // the next step's first operands are in flight.  Carried from step to step: the un-normalised row w and
          // its sum; beta_init is the same thing with w = 1.0f (sum = K by sequential adds, HMM.cpp:887-897).
          float bsum = 0.f;
#pragma unroll
          for (int k = 0; k < K; ++k) {
            w[k] = (!kGhost<KT> || k < p.K) ? 1.0f : 0.f;
            bsum = bsum + w[k];
          }
          const float bsumInit = bsum;
          // DUAL: the lanes whose own window ends at site q start from beta = 1 there (HMM.cpp:887-897)
          auto openBeta = [&](const int q) {
            if constexpr (DUAL) {
              if (q == toA - 1 || q == toB - 1) {
                const bool sel = myTo - 1 == q;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                  w[k] = sel ? ((!kGhost<KT> || k < p.K) ? 1.0f : 0.f) : w[k];
                }
                bsum = sel ? bsumInit : bsum;
              }
            }
          };
          bool stored = false;
          for (int pos = to - 2; pos >= from; --pos) {
            const int q = pos + 1;
            openBeta(q);
            waitEmisRows(stored); // the rows of site q have landed
            __builtin_amdgcn_wave_barrier();
            if (pos - 1 >= from) {
              stageEmis(q - 1); // into the slot of site q + 1, whose step is over
            }
            const int row = stepRowOf(q);
            const int c = obsClass(q);
            const float4* e = &emisLds[q & 1][c * E4];
            const cfloat_p rs = rowSetOf<KT>(tabs, row);
            BetaOps<KT> ops;
            beta_issue_pk<KT>(ops, rs, e);
            FSMC_END(cycW, 0);
            scale_pk<KT, KA>(b, w, bsum); // beta of site q, final
            stored = afterBeta(q);
            FSMC_END(cycW, 1);
            bsum = beta_core_pk<KT, KA, kGhost<KT>>(b, w, ops, rs, e, tabs.ghostMask, cycW);
          }
          openBeta(from);
          scale_pk<KT, KA>(b, w, bsum);
          afterBeta(from);
        } else {
          for (int pos = to - 2; pos >= from; --pos) {
            const int q = pos + 1;
            waitVm0();
            __builtin_amdgcn_wave_barrier();
            if (pos - 1 >= from) {
              stageEmis(q - 1);
            }
            const int row = stepRowOf(q);
            const int c = obsClass(q);
            beta_step<KT, KA, true, kSeqSyncLoads<SEQ, KT, MODE>>(K, b, w, tabs, row, &emisLds[q & 1][c * E4], cycW);
            afterBeta(pos);
          }
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
#if defined(FSMC_DIAG_NOSMEM) || defined(FSMC_DIAG_NOEMIS) || defined(FSMC_DIAG_SAMEROW)
      if (s0 >= 0) { // garbage posteriors would flood the record buffer
        return;
      }
#endif
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
    auto fetchBeta = [&](const float4* row) { // row: wave-uniform address of the stored vector
      const gchar_p base = uniformPtr(row);
#pragma unroll
      for (int k4 = 0; k4 < K4; ++k4) {
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_global_load_lds(rowSlot(base, k4, laneOff), &betaLds[k4 * kWave], 16, 0, 2 /* nt */);
#endif
      }
    };

    for (int j = 0; j < (nChunks > 0 ? nChunks : 0); ++j) {
      {
        // (two thirds of the work are left when pass A starts: levels 2, 1, 0 as the chunks go by)
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
      // the rows of this chunk: kept by pass B (a resident chunk), or rebuilt below into the wave's chunk buffer
      const bool resident = kResidentBuilt && !single && j < nResident;
      float4* const cbuf = resident ? resBase + (size_t)j * p.chunkRows * vecF4 : chunkbuf;
      if (!single && !resident) {
        // park the carried alpha while the chunk's betas are rebuilt
        if (j > 0) {
          store_vec<KT, KA>(K, saveA, laneOff, a);
        }
        auto storeRow = [&](const int site, const float (&row)[KA]) -> bool { // true if the row went out
          const int rel = site - lo;
          if (!HALF || (rel & 1) || site == hi - 1) {
            store_vec<KT, KA>(K, chunkbuf + slotOf(rel) * vecF4, laneOff, row);
            return true;
          }
          return false;
        };
        if constexpr (kRotate) {
          // same rotation as pass B: (w, bsum) is the pending un-normalised row; a checkpoint is loaded as a row with
          // sum 1.0f (w * (1.0f / 1.0f) is an exact copy) and belongs to the next chunk (not stored here)
          float b[KA];
          float bsum = 0.f;
          int pos;
          if (hi == to) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
              w[k] = (!kGhost<KT> || k < p.K) ? 1.0f : 0.f;
              bsum = bsum + w[k];
            }
            pos = to - 2;
          } else {
            load_vec<KT, KA>(K, ckpt + (size_t)(j + 1) * vecF4, laneOff, w);
            bsum = 1.0f;
            pos = hi - 1;
          }
          if (pos >= lo) {
            stageEmis(pos + 1);
          }
          // DUAL: the lanes whose own window ends at site q start from beta = 1 there, as in pass B (HMM.cpp:887-897)
          auto reopenBeta = [&](const int q) {
            if constexpr (DUAL) {
              if (q == toA - 1 || q == toB - 1) {
                const bool sel = myTo - 1 == q;
                float ones = 0.f;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                  const float one = (!kGhost<KT> || k < p.K) ? 1.0f : 0.f;
                  w[k] = sel ? one : w[k];
                  ones = ones + one;
                }
                bsum = sel ? ones : bsum;
              }
            }
          };
          bool stored = false;
          for (; pos >= lo; --pos) {
            const int q = pos + 1;
            reopenBeta(q);
            waitEmisRows(stored); // the rows of site q have landed
            __builtin_amdgcn_wave_barrier();
            if (pos - 1 >= lo) {
              stageEmis(q - 1); // into the slot of site q + 1, whose step is over
            }
            const int row = stepRowOf(q);
            const int c = obsClass(q);
            const float4* e = &emisLds[q & 1][c * E4];
            const cfloat_p rs = rowSetOf<KT>(tabs, row);
            BetaOps<KT> ops;
            beta_issue_pk<KT>(ops, rs, e);
            FSMC_END(cycW, 0);
            scale_pk<KT, KA>(b, w, bsum); // beta of site q, final
            stored = q < hi && storeRow(q, b);
            FSMC_END(cycW, 1);
            bsum = beta_core_pk<KT, KA, kGhost<KT>>(b, w, ops, rs, e, tabs.ghostMask, cycW);
          }
          reopenBeta(pos + 1);
          scale_pk<KT, KA>(b, w, bsum);
          if (pos + 1 < hi) {
            storeRow(pos + 1, b);
          }
        } else {
          float b[KA];
          int pos;
          if (hi == to) {
            beta_init<KT, KA>(K, p.K, b);
            if constexpr (SEQ) {
              if (to - 1 > from) {
                betaGapStep(b, to - 1, prefetchEmis(to - 1));
              }
            }
            store_vec<KT, KA>(K, chunkbuf + slotOf(to - 1 - lo) * vecF4, laneOff, b);
            pos = to - 2;
          } else {
            load_vec<KT, KA>(K, ckpt + (size_t)(j + 1) * vecF4, laneOff, b);
            pos = hi - 1;
            if constexpr (SEQ) {
              commitEmis(hi, prefetchEmis(hi)); // the checkpoint is the stored vector of site hi: its rows next
            }
          }
          if constexpr (SEQ) {
            for (; pos >= lo; --pos) {
              betaSeqStep(b, pos);
              store_vec<KT, KA>(K, chunkbuf + (size_t)(pos - lo) * vecF4, laneOff, b);
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
              const int row = stepRowOf(q);
              const int c = obsClass(q);
              beta_step<KT, KA, true, kSeqSyncLoads<SEQ, KT, MODE>>(K, b, w, tabs, row, &emisLds[q & 1][c * E4], cycW);
              storeRow(pos, b);
            }
          }
        }
        if (j > 0) {
          load_vec<KT, KA>(K, saveA, laneOff, a);
        }
      }

      FSMC_END(cycW, 13);
      FSMC_STAMP(cycR);
      // the wave's own stores of this chunk's betas must have landed before the DMA reads them back
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      waitVm0();
      fetchBeta(cbuf);
      EmisRegs ev;
      if constexpr (SEQ) {
        ev = prefetchEmis(lo);
        commitEmis(lo, ev); // later sites are staged by the half-step of the site before them
      } else {
        // Array mode: the ring holds the rows of sites pos and pos + 1; after the combine of site pos its slot is
        // free and takes the rows of site pos + 2, which the vmcnt(0) wait of site pos + 1 covers.
        stageEmis(lo);
        if (lo + 1 < hi) {
          stageEmis(lo + 1);
        }
      }
      // table rows of the steps into the next sites, requested two sites ahead (a scalar load waited for at its use
      // costs the wave a round trip to L2 at every site)
      // every request so far (the first beta row, the first sites' emission rows) has landed: inside the loop the
      // emission registers are then only ever read behind one of its own vmcnt(0) waits, and the compiler adds none
      waitVm0();
      for (int pos = lo; pos < hi; ++pos) {
        // HALF: this site's beta row was not stored -- it is recomputed below from the row of site pos+1
        const bool rec = HALF && ((pos - lo) & 1) == 0 && pos + 1 < hi;

        if constexpr (SEQ) {
          if (pos < to - 1) {
            ev = prefetchEmis(pos + 1);
          }
        }
        const int c = obsClass(pos);
        const float4* e = &emisLds[pos & 1][c * E4];
        FSMC_END(cycW, 4);
        if (__builtin_expect(pos == from, 0)) {
          alpha_init<KT, KA>(K, a, tPi, e);
        } else {
          alpha_step<KT, KA, true, kSeqSyncLoads<SEQ, KT, MODE>>(K, a, w, tabs, stepRowOf(pos), e, cycW);
          if constexpr (DUAL) {
            // the lanes whose own window opens here start from pi * emission (HMM.cpp:736-747)
            if (pos == fromA || pos == fromB) {
              alpha_init<KT, KA>(K, w, tPi, e);
              const bool sel = myFrom == pos;
#pragma unroll
              for (int k = 0; k < K; ++k) {
                a[k] = sel ? w[k] : a[k];
              }
            }
          }
        }
        if constexpr (SEQ) {
          // what the reference's alpha buffer holds for this site: alpha after the un-normalised half-step
          // across the gap to the next site (HMM.cpp:764-767); the last site of the window keeps its alpha
          if (pos < to - 1) {
            commitEmis(pos + 1, ev);
            const int row = tRowGapF[pos + 1];
            alpha_step<KT, KA, false, kSeqSyncLoads<SEQ, KT, MODE>>(K, a, w, tabs, row, &emisLds[(pos + 1) & 1][3 * E4],
                                            cycW);
          }
        }

        // combine with beta of this site (landed in LDS) and normalise (HMM.cpp:672-691)
        waitVm0();
        __builtin_amdgcn_wave_barrier();
        FSMC_END(cycW, 8);
        float sumq = 0.f;
        if (rec) {
          // the landing zone holds beta of site pos+1: one beta step back gives this site's row (the same
          // operations, on the same bits, as the pass that stored its neighbours), combined from registers
          float b[KA];
          const int q = pos + 1;
          const int cq1 = obsClass(q);
          const int rowq = stepRowOf(q);
          const float4* eq = &emisLds[q & 1][cq1 * E4];
          auto readLanded = [&]() {
#pragma unroll
            for (int k4 = 0; k4 < K4; ++k4) {
              const float4 o = betaLds[k4 * kWave + lane];
              b[4 * k4] = o.x;
              if (4 * k4 + 1 < K) b[4 * k4 + 1] = o.y;
              if (4 * k4 + 2 < K) b[4 * k4 + 2] = o.z;
              if (4 * k4 + 3 < K) b[4 * k4 + 3] = o.w;
            }
          };
          {
            const cfloat_p rs = rowSetOf<KT>(tabs, rowq);
            BetaOps<KT> ops;
            beta_issue_pk<KT>(ops, rs, eq); // in flight while the landed row moves from LDS to registers
            readLanded();
            FSMC_END(cycW, 9);
            const float bsum = beta_core_pk<KT, KA, kGhost<KT>>(b, w, ops, rs, eq, tabs.ghostMask, cycW);
            scale_pk<KT, KA>(b, w, bsum);
            if constexpr (DUAL) {
              // the lanes whose own window ends at this site start from beta = 1 here (HMM.cpp:887-897), exactly what
              // pass B gave them (and did not store: the row of an even offset): 1.0f * (1.0f / sum of K ones)
              if (__builtin_expect(pos == toA - 1 || pos == toB - 1, 0)) {
                float ones = 0.f;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                  ones = ones + ((!kGhost<KT> || k < p.K) ? 1.0f : 0.f);
                }
                const float c1 = 1.0f / ones;
                const bool sel = myTo - 1 == pos;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                  const float init = ((!kGhost<KT> || k < p.K) ? 1.0f : 0.f) * c1;
                  b[k] = sel ? init : b[k];
                }
              }
            }
          }
          FSMC_END(cycW, 9);
#pragma unroll
          for (int k = 0; k < K; k += 2) {
            if (k + 1 < K) { // products two states at a time, the sum in state order
              const f32x2 av = {a[k], a[k + 1]};
              const f32x2 bv = {b[k], b[k + 1]};
              const f32x2 q = pmul(av, bv);
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
              if (k + 1 < K) {
                const f32x2 av = {a[k], a[k + 1]};
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
        FSMC_END(cycW, 10);
        // every read of the landing zone has returned: request the next site's beta row
        waitLgkm0();
        if (MODE != kModeSums && !rec && pos + 1 < hi) {
          fetchBeta(cbuf + slotOf(pos + 1 - lo) * vecF4);
        }
        if constexpr (!SEQ && MODE != kModeSums) {
          if (pos + 2 < hi) {
            stageEmis(pos + 2); // this site's ring slot is free: its alpha step (and beta recompute) are over
          }
        }

        FSMC_END(cycW, 11);
        if (MODE == kModePerPair) {
          // HMM::writePerPairOutput (HMM.cpp:1378-1409): mean = sum_k post*E[t_k] (k ascending from 0.f),
          // MAP = first strictly larger posterior
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

        if (MODE == kModeSums) {
          // HMM::augmentSumOverPairs (HMM.cpp:1052-1081): per site and state, the batch's posteriors are summed
          // over pairs in batch order (local fp32 sum from 0.f), then added to the accumulator.  The K x 64 tile
          // is transposed through LDS (row stride 65 floats: conflict-free both ways); lane j then owns state j.
          float* const tile = reinterpret_cast<float*>(betaLds);
          // this site's ring slot: its rows are no longer needed
          unsigned char* const cls = reinterpret_cast<unsigned char*>(&emisLds[pos & 1][0]);
#pragma unroll
          for (int k = 0; k < K; ++k) {
            tile[k * 65 + lane] = w[k] * cq;
          }
          if (p.flags & FSMC_WANT_MAJOR_MINOR_SUMS) {
            cls[lane] = (unsigned char)c;
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          // two states a lane in one walk over the pairs (K = 69: the five states beyond the first 64 ride along in lanes
          // 0..4 instead of costing a second walk); every state's sum still adds its pairs in batch order
          for (int kb = 0; kb < Kreal; kb += 2 * kWave) {
            const int kk0 = kb + lane, kk1 = kb + kWave + lane;
            const bool h0 = kk0 < Kreal, h1 = kk1 < Kreal;
            float* const acc0 = p.sums + (size_t)blockIdx.x * p.sumsSlot + (size_t)pos * Kreal + (h0 ? kk0 : 0);
            float* const acc1 = p.sums + (size_t)blockIdx.x * p.sumsSlot + (size_t)pos * Kreal + (h1 ? kk1 : 0);
            float s[2] = {0.f, 0.f}, s00[2] = {0.f, 0.f}, s01[2] = {0.f, 0.f}, s11[2] = {0.f, 0.f};
            if (round > 0) { // a later group of the batch: the running sums of the pairs before (this wave wrote them)
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
            // The walk over the batch's pairs, sixteen at a time: the tile values of sixteen pairs are read together
            // (one wait), then added one after the other -- the reference's order of additions (HMM.cpp:1054-1073).
            // As a loop of one pair a turn (rounds 1-3) every turn waited for its own two LDS reads and took a branch:
            // ~70 cycles a pair on a wave that runs alone.  The 00 / 01 / 11 split adds +0.f to the two sums a pair
            // does not belong to (x + 0.f == x for the non-negative sums) instead of branching on the pair's class.
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
              int v = 0;
              for (; v + kWalk <= nPairsInGroup; v += kWalk) {
                float q0[kWalk], q1[kWalk];
                int cv[kWalk];
#pragma unroll
                for (int i = 0; i < kWalk; ++i) {
                  q0[i] = tile[t0 + v + i];
                  q1[i] = tile[t1 + v + i];
                  cv[i] = SPLIT ? (int)cls[v + i] : 0;
                }
#pragma unroll
                for (int i = 0; i < kWalk; ++i) {
                  add(q0[i], q1[i], cv[i]);
                }
              }
              for (; v < nPairsInGroup; ++v) {
                add(tile[t0 + v], tile[t1 + v], SPLIT ? (int)cls[v] : 0);
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
          if (pos + 1 < hi) {
            fetchBeta(cbuf + slotOf(pos + 1 - lo) * vecF4);
          }
          if constexpr (!SEQ) {
            if (pos + 2 < hi) {
              stageEmis(pos + 2);
            }
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

        if (MODE == kModeIbd) {
          if (pos >= scanFrom) {
            // posterior of the states the scan needs
            // (the states beyond the scan's only feed the per-state sums of open segments: scaled there, by the
            //  lanes that are inside a segment)
            // Sum over the states below the threshold, k ascending from 0.f (HMM.cpp:1207-1224).  The loop leaves at
            // the first block of four states beyond the threshold -- one taken branch per site; a guard around every
            // block was seventeen of them (a taken branch costs this in-order wave ~100 cycles: the scan was 14 % of
            // the kernel).  Fully unrolled, so the register arrays stay statically indexed.  Inside the last block
            // the states beyond the threshold add +0.f, which leaves the (non-negative) sum unchanged.
            const unsigned nPost = p.stateThr;
            float s = 0.f;
            scanBlocks<KA, KT, K4A>(w, s, cq, launderScalar(nPost));
            int level = s >= p.thr[0] ? 0 : s >= p.thr[1] ? 1 : s >= p.thr[2] ? 2 : s >= p.thr[3] ? 3 : 4;
            if constexpr (DUAL) {
              if (!(pos >= mySF && pos < myST)) {
                level = 4; // outside this lane's own scan window
              }
            }
            // a change of level (or a drop below every threshold) closes the open segment at pos-1
            if (__builtin_expect(valid && cur != 4 && level != cur, 0)) {
              emit(segStart, pos - 1);
            }
            const bool opening = level != 4 && level != cur;
            if constexpr (TRACK) {
              // per-state posterior sums of the open segment (sum_posterior_per_state, HMM.cpp:1212-1229): kept
              // in the wave's workspace, touched only by the lanes that are inside a segment at this site
              if (level != 4) {
                // four blocks (sixteen states) per round trip: the loads of a round go out together.  A lane that
                // opens a segment at this site starts from zero: it clears its column first (once per segment), so
                // that the accumulation needs no select per state; 0.f + x is x, and a wave's own store to an
                // address is what its next load from it returns.  The products and sums go two states an instruction
                // (the same IEEE operations): this block is executed at every site that has ANY lane of the wave
                // inside a segment.
                // A member that runs ONE wave per SIMD has nobody to hide a round trip behind and the accumulation
                // registers to park other values in: all its blocks' loads go out together, one round trip a site
                // (K = 80 on the C2 shape: 0.73 -> 0.77, K = 100: 0.69 -> 0.72).
                constexpr int kG = 4;
                constexpr bool kOneTrip = minWavesPerSimd(KT) == 1;
                const gchar_p spsBase = uniformPtr(saveS); // scalar base + lane offset + immediate
                // (the two thresholds as values the compiler cannot prove loop-invariant, like the scan's: it
                //  otherwise hoists the compare of EVERY block out of the site loop as a lane mask in a scalar pair,
                //  spills them all and reloads two lanes of a spill register per block and site)
                const unsigned nAgeL = launderScalar(p.ageThr), nPostL = launderScalar(nPost);
                auto rounds = [&](auto inLds) {
                  constexpr bool LDS = decltype(inLds)::value;
                  auto loadBlock = [&](const int k4) -> float4 {
                    if constexpr (LDS) {
                      return spsDyn[k4 * kWave + lane];
                    } else {
                      const f32x4 t = *rowSlot(spsBase, k4, laneOff);
                      return make_float4(t.x, t.y, t.z, t.w);
                    }
                  };
                  // sv = the block's sums so far; adds this site's posteriors and writes the block back
                  auto addBlock = [&](const int k4, float4 sv) {
                    // blocks the scan already normalised are taken as they are (x * 1.0f is exact); states at or
                    // beyond the age threshold are never read back (segment_ages stops there)
                    const float sc = ((unsigned)(4 * k4) < nPostL) ? 1.0f : cq;
                    if (4 * k4 + 3 < K) {
                      const f32x2 scv = {sc, sc};
                      const f32x2 w01 = {w[4 * k4], w[4 * k4 + 1]}, w23 = {w[4 * k4 + 2], w[4 * k4 + 3]};
                      const f32x2 s01 = {sv.x, sv.y}, s23 = {sv.z, sv.w};
                      const f32x2 r01 = padd(s01, pmul(w01, scv)), r23 = padd(s23, pmul(w23, scv));
                      sv = make_float4(r01.x, r01.y, r23.x, r23.y);
                    } else {
                      sv.x = sv.x + w[4 * k4] * sc;
                      if (4 * k4 + 1 < K) sv.y = sv.y + w[4 * k4 + 1] * sc;
                      if (4 * k4 + 2 < K) sv.z = sv.z + w[4 * k4 + 2] * sc;
                      if (4 * k4 + 3 < K) sv.w = sv.w + w[4 * k4 + 3] * sc;
                    }
                    if constexpr (LDS) {
                      spsDyn[k4 * kWave + lane] = sv;
                    } else {
                      const f32x4 t = {sv.x, sv.y, sv.z, sv.w};
                      *rowSlot(spsBase, k4, laneOff) = t;
                    }
                  };
                  if (__builtin_expect(opening, 0)) {
#pragma unroll
                    for (int k4 = 0; k4 < K4; ++k4) {
                      if ((unsigned)(4 * k4) >= nAgeL) {
                        break;
                      }
                      if constexpr (LDS) {
                        spsDyn[k4 * kWave + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
                      } else {
                        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                        *rowSlot(spsBase, k4, laneOff) = z;
                      }
                    }
                  }
                  if constexpr (kOneTrip && !LDS) {
                    float4 sv[K4];
#pragma unroll
                    for (int g4 = 0; g4 < K4; g4 += kG) {
                      if ((unsigned)(4 * g4) < nAgeL) { // (groups of four blocks under the age threshold)
#pragma unroll
                        for (int j = 0; j < kG; ++j) {
                          if (g4 + j < K4) {
                            sv[g4 + j] = loadBlock(g4 + j);
                          }
                        }
                      }
                    }
#pragma unroll
                    for (int g4 = 0; g4 < K4; g4 += kG) {
                      if ((unsigned)(4 * g4) < nAgeL) {
#pragma unroll
                        for (int j = 0; j < kG; ++j) {
                          if (g4 + j < K4) {
                            addBlock(g4 + j, sv[g4 + j]);
                          }
                        }
                      }
                    }
                  } else {
                    // four blocks (sixteen states) per round trip
#pragma unroll
                    for (int g4 = 0; g4 < K4; g4 += kG) {
                      if ((unsigned)(4 * g4) >= nAgeL) {
                        break;
                      }
                      float4 sv[kG];
#pragma unroll
                      for (int j = 0; j < kG; ++j) {
                        if (g4 + j < K4) {
                          sv[j] = loadBlock(g4 + j);
                        }
                      }
#pragma unroll
                      for (int j = 0; j < kG; ++j) {
                        if (g4 + j < K4) {
                          addBlock(g4 + j, sv[j]);
                        }
                      }
                    }
                  }
                };
                if constexpr (kSpsLdsBuilt<MODE, SEQ, DUAL>) {
                  if (spsInLds) { // (uniform over the launch)
                    rounds(std::true_type{});
                  } else {
                    rounds(std::false_type{});
                  }
                } else {
                  rounds(std::false_type{});
                }
              }
            }
            acc = (level == 4) ? 0.f : (opening ? s : acc + s);
            if (opening) {
              segStart = pos;
            }
            cur = level;
            if constexpr (DUAL) {
              if (__builtin_expect(pos == stA - 1 || pos == stB - 1, 0)) {
                if (pos == myST - 1) { // the last site of this lane's scan window closes its open segment
                  if (valid && cur != 4) {
                    emit(segStart, pos);
                  }
                  cur = 4;
                }
              }
            } else if (__builtin_expect(pos == aEnd - 1, 0)) {
              if (valid && cur != 4) {
                emit(segStart, pos);
              }
            }
          }
        }
        FSMC_END(cycW, 12);
      }
      FSMC_STAMP(cycA);
    }
#if defined(FSMC_REGION_STAMPS)
    if (lane == 0 && p.phaseCycles) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        atomicAdd(&p.phaseCycles[8 + r], (unsigned long long)cycW.acc[r]);
      }
    }
#endif
#if defined(FSMC_DIAG_GROUP_TIMES)
    // diagnostic builds (tools/build_variant.sh <name> -DFSMC_DIAG_GROUP_TIMES; tools/analyse_group_times.py): one marker
    // record per group -- start = -1, end = the wave's slot, prob / post_mean = when the group ended / began (ms of the
    // 100 MHz clock), map = HW_ID and XCC_ID -- among the IBD records.  How the SIMD-sharing of the waves was found.
    if (MODE == kModeIbd && lane == 0) {
      const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
      const unsigned idx = atomicAdd(&p.counters[1], 1u);
      if (idx < p.recCap) {
        fsmc_ibd_record r;
        r.pair = pairIdx;
        r.start = -1;
        r.end = (int)blockIdx.x;
        r.prob = (float)((double)(t1 & 0xFFFFFFFFull) / 1e5);
        r.post_mean = (float)((double)(diagT0 & 0xFFFFFFFFull) / 1e5);
        r.map = __int_as_float((int)(__builtin_amdgcn_s_getreg(63492) & 0xFFFFu) |
                               (int)((__builtin_amdgcn_s_getreg(63508) & 0xFu) << 16));
        p.recs[idx] = r;
      }
    }
#endif
#if defined(FSMC_PHASE_STAMPS)
    if (lane == 0 && p.phaseCycles) {
      atomicAdd(&p.phaseCycles[0], (unsigned long long)cycB);
      atomicAdd(&p.phaseCycles[1], (unsigned long long)cycR);
      atomicAdd(&p.phaseCycles[2], (unsigned long long)cycA);
      atomicAdd(&p.phaseCycles[3], 1ull);
      atomicAdd(&p.phaseCycles[4], (unsigned long long)cycW.waitCycles);
    }
#endif
  }
}


} // namespace fsmc
