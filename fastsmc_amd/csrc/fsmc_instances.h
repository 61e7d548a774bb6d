// fsmc_instances.h -- the kernel instantiations of libfastsmc_hip.so and where each is compiled.
//
// The decode kernels are templates (fsmc_kernels.h, fsmc_kernels_w2.h); every instantiation is a fully unrolled kernel
// of several thousand instructions, so they are compiled in separate translation units, in parallel: fsmc_inst.hip is
// built once per member (-DFSMC_INSTANCE_KT=<n> / -DFSMC_INSTANCE_W2=<n>) and defines the instantiations of that
// member; fsmc_capi.hip only sees the declarations below and picks a function pointer.
//
// Lane-per-pair family (decode_kernel<KT, MODE, TRACK, SEQ, HALF, DUAL>), one wave per group:
//   KT = 69               the 69-state models of the reference's decoding-quantities files, no padding
//   KT = 16, 32, 48, 64, 80   every other model with K <= 80: padded with ghost states to the next member
//   KT = 96, 112, 128     80 < K <= 128: the same kernel with one wave per SIMD (512 registers a lane)
//   exact members (FSMC_EXACT_KT, default 50 and 100; any K that is not a multiple of 16 may be listed at build time:
//   `FSMC_EXACT_MEMBERS="50 100 75" python -m fastsmc_amd.build`): a model of exactly that many states runs without
//   ghost states, masks and run-time state counts -- what the 69-state member is to the reference's own files
// Two waves per window (decode_kernel_bidir<KT, MODE>, fsmc_kernels_bidir.h): the same members' dump / per-pair / sums
//   consumers in array mode for launches that leave half the chip empty -- alpha from the window's start in one wave while
//   beta comes from its end in another.
// Wave-group kernel (decode_kernel_w2<KH, MODE, TRACK, SEQ, NW>), lane = pair and NW waves per group of KH states each:
//   128 < K <= 256: four waves of 48 or 64 states; 256 < K <= 320: four waves of 80; 320 < K <= 512: six, seven or eight
//   waves of 64; 512 < K <= 1024: eight waves of 80, 96 or 128 (no landing zones).
// Any-K kernel (decode_kernel_any<MODE, TRACK, SEQ>, fsmc_kernels_any.h): K > 1024, a pair's K-vectors in the workspace
//   instead of registers -- correct, not fast; small enough to be instantiated where it is picked (fsmc_capi.hip).
// (The runtime-K instantiation KT = 0 and the four-lanes-per-pair kernel of earlier builds are gone: every model of at
//  most 256 states, in every mode, runs one of the two families above.)
#pragma once

#include "fsmc_kernels.h"
#include "fsmc_kernels_any.h"
#include "fsmc_kernels_bidir.h"
#include "fsmc_kernels_w2.h"

namespace fsmc
{

// Exact (ghost-free) members beside the 69-state one: Y(K) for every K listed (none a multiple of 16, each <= 128)
#ifndef FSMC_EXACT_KT
#define FSMC_EXACT_KT(Y) Y(50) Y(100)
#endif
#define FSMC_IS_EXACT_TERM(KTX) || KT == KTX
constexpr bool exactMember(const int KT)
{
  return KT == 69 FSMC_EXACT_KT(FSMC_IS_EXACT_TERM);
}

// The sums over pairs with beta stride 2: every member that has stride 2 EXCEPT the exact 50-state one, whose
// instantiation the compiler builds with an operand block spilled while its scalar load is in flight
// (tools/check_inflight_sgprs.py: v_writelane of the load's destination before the wait) -- it keeps stride 1.
constexpr bool halfSumsBuilt(const int KT);
// beta stride 2 needs three K-vectors in a lane's registers: built for the members it fits
constexpr bool halfBuilt(const int KT)
{
  return KT == 16 || KT == 32 || KT == 48 || KT == 64 || KT == 80 || KT == 96 || KT == 112 || KT == 128 || exactMember(KT);
}

constexpr bool halfSumsBuilt(const int KT)
{
  return halfBuilt(KT) && KT != 50;
}

#define FSMC_KT_KERNELS(X, KT)                                                                                         \
  X(KT, kModeIbd, true, false, false)                                                                                  \
  X(KT, kModeIbd, false, false, false)                                                                                 \
  X(KT, kModeIbd, true, true, false)                                                                                   \
  X(KT, kModeIbd, false, true, false)                                                                                  \
  X(KT, kModeDump, false, false, false)                                                                                \
  X(KT, kModeDump, false, true, false)                                                                                 \
  X(KT, kModePerPair, false, false, false)                                                                             \
  X(KT, kModePerPair, false, true, false)                                                                              \
  X(KT, kModeSums, false, false, false)                                                                                \
  X(KT, kModeSums, false, true, false)
#define FSMC_KT_HALF_KERNELS(X, KT)                                                                                    \
  X(KT, kModeIbd, true, false, true)                                                                                   \
  X(KT, kModeIbd, false, false, true)
// ... and the sums over pairs with beta stride 2 (round 5), where the instantiation passes the in-flight check
#define FSMC_KT_HALF_SUMS_KERNELS(X, KT) X(KT, kModeSums, false, false, true)

#define FSMC_DECLARE_KT(KT, MODE, TRACK, SEQ, HALF)                                                                    \
  extern template __global__ void decode_kernel<KT, MODE, TRACK, SEQ, HALF>(const KParams);
#define FSMC_DEFINE_KT(KT, MODE, TRACK, SEQ, HALF)                                                                     \
  template __global__ void decode_kernel<KT, MODE, TRACK, SEQ, HALF>(const KParams);
// two waves per window (fsmc_kernels_bidir.h): the consumers without state across sites, array mode, small launches
#define FSMC_KT_BIDIR_KERNELS(X, KT) X(KT, kModeDump) X(KT, kModePerPair) X(KT, kModeSums)
#define FSMC_DECLARE_KT_BIDIR(KT, MODE) extern template __global__ void decode_kernel_bidir<KT, MODE>(const KParams);
#define FSMC_DEFINE_KT_BIDIR(KT, MODE) template __global__ void decode_kernel_bidir<KT, MODE>(const KParams);
// two half-groups per wavefront (hashing-mode work lists): array-mode IBD decode, beta stride 1
#define FSMC_DECLARE_KT_DUAL(KT)                                                                                       \
  extern template __global__ void decode_kernel<KT, kModeIbd, true, false, false, true>(const KParams);              \
  extern template __global__ void decode_kernel<KT, kModeIbd, false, false, false, true>(const KParams);
#define FSMC_DEFINE_KT_DUAL(KT)                                                                                        \
  template __global__ void decode_kernel<KT, kModeIbd, true, false, false, true>(const KParams);                     \
  template __global__ void decode_kernel<KT, kModeIbd, false, false, false, true>(const KParams);
// ... with beta stride 2 (the members beta stride 2 is built for)
#define FSMC_DECLARE_KT_DUAL_HALF(KT)                                                                                  \
  extern template __global__ void decode_kernel<KT, kModeIbd, true, false, true, true>(const KParams);               \
  extern template __global__ void decode_kernel<KT, kModeIbd, false, false, true, true>(const KParams);
#define FSMC_DEFINE_KT_DUAL_HALF(KT)                                                                                   \
  template __global__ void decode_kernel<KT, kModeIbd, true, false, true, true>(const KParams);                      \
  template __global__ void decode_kernel<KT, kModeIbd, false, false, true, true>(const KParams);
// NW waves per group of KH states each, lane = pair (fsmc_kernels_w2.h): 128 < K <= 512
#define FSMC_W2_MODE_KERNELS(X, KH, NW, SEQ)                                                                           \
  X(KH, kModeIbd, true, SEQ, NW)                                                                                       \
  X(KH, kModeIbd, false, SEQ, NW)                                                                                      \
  X(KH, kModeDump, false, SEQ, NW)                                                                                     \
  X(KH, kModeSums, false, SEQ, NW)                                                                                     \
  X(KH, kModePerPair, false, SEQ, NW)
// (array mode and sequence mode: one translation unit, or -- -DFSMC_INSTANCE_SEQ=0|1, the widest members -- one each)
#define FSMC_W2_KERNELS(X, KH, NW) FSMC_W2_MODE_KERNELS(X, KH, NW, false) FSMC_W2_MODE_KERNELS(X, KH, NW, true)
#define FSMC_DECLARE_W2(KH, MODE, TRACK, SEQ, NW)                                                                      \
  extern template __global__ void decode_kernel_w2<KH, MODE, TRACK, SEQ, NW>(const KParams);
#define FSMC_DEFINE_W2(KH, MODE, TRACK, SEQ, NW)                                                                       \
  template __global__ void decode_kernel_w2<KH, MODE, TRACK, SEQ, NW>(const KParams);

// every member of the library (build.py compiles fsmc_inst.hip once for each entry of these two lists)
#define FSMC_ALL_KT(Y) Y(16) Y(32) Y(48) Y(64) Y(69) Y(80) Y(96) Y(112) Y(128)
#define FSMC_ALL_W2(Y) Y(48, 4) Y(64, 4) Y(80, 4) Y(64, 6) Y(64, 7) Y(64, 8) Y(80, 8) Y(96, 8) Y(128, 8)

#if !defined(FSMC_INSTANCE_KT) && !defined(FSMC_INSTANCE_W2)
#define FSMC_DECLARE_MEMBER(KT) FSMC_KT_KERNELS(FSMC_DECLARE_KT, KT) FSMC_KT_BIDIR_KERNELS(FSMC_DECLARE_KT_BIDIR, KT)
FSMC_ALL_KT(FSMC_DECLARE_MEMBER)
#define FSMC_DECLARE_EXACT_MEMBER(KT)                                                                                   \
  static_assert(KT % 16 != 0 && KT <= 128, "an exact member is not a multiple of 16 states and has at most 128");       \
  FSMC_KT_KERNELS(FSMC_DECLARE_KT, KT)                                                                                  \
  FSMC_KT_BIDIR_KERNELS(FSMC_DECLARE_KT_BIDIR, KT)                                                                      \
  FSMC_KT_HALF_KERNELS(FSMC_DECLARE_KT, KT)                                                                             \
  FSMC_KT_HALF_SUMS_KERNELS(FSMC_DECLARE_KT, KT)                                                                        \
  FSMC_DECLARE_KT_DUAL_HALF(KT)                                                                                         \
  FSMC_DECLARE_KT_DUAL(KT)
FSMC_EXACT_KT(FSMC_DECLARE_EXACT_MEMBER)
FSMC_KT_HALF_KERNELS(FSMC_DECLARE_KT, 16) FSMC_KT_HALF_SUMS_KERNELS(FSMC_DECLARE_KT, 16)
FSMC_KT_HALF_KERNELS(FSMC_DECLARE_KT, 32) FSMC_KT_HALF_SUMS_KERNELS(FSMC_DECLARE_KT, 32)
FSMC_KT_HALF_KERNELS(FSMC_DECLARE_KT, 48) FSMC_KT_HALF_SUMS_KERNELS(FSMC_DECLARE_KT, 48)
FSMC_KT_HALF_KERNELS(FSMC_DECLARE_KT, 64) FSMC_KT_HALF_SUMS_KERNELS(FSMC_DECLARE_KT, 64)
FSMC_KT_HALF_KERNELS(FSMC_DECLARE_KT, 69) FSMC_KT_HALF_SUMS_KERNELS(FSMC_DECLARE_KT, 69)
FSMC_KT_HALF_KERNELS(FSMC_DECLARE_KT, 80) FSMC_KT_HALF_SUMS_KERNELS(FSMC_DECLARE_KT, 80)
FSMC_KT_HALF_KERNELS(FSMC_DECLARE_KT, 96) FSMC_KT_HALF_SUMS_KERNELS(FSMC_DECLARE_KT, 96)
FSMC_KT_HALF_KERNELS(FSMC_DECLARE_KT, 112) FSMC_KT_HALF_SUMS_KERNELS(FSMC_DECLARE_KT, 112)
FSMC_KT_HALF_KERNELS(FSMC_DECLARE_KT, 128) FSMC_KT_HALF_SUMS_KERNELS(FSMC_DECLARE_KT, 128)
FSMC_DECLARE_KT_DUAL_HALF(16)
FSMC_DECLARE_KT_DUAL_HALF(32)
FSMC_DECLARE_KT_DUAL_HALF(48)
FSMC_DECLARE_KT_DUAL_HALF(64)
FSMC_DECLARE_KT_DUAL_HALF(69)
FSMC_DECLARE_KT_DUAL_HALF(80)
FSMC_DECLARE_KT_DUAL_HALF(96)
FSMC_DECLARE_KT_DUAL_HALF(112)
FSMC_DECLARE_KT_DUAL_HALF(128)
FSMC_DECLARE_KT_DUAL(16)
FSMC_DECLARE_KT_DUAL(32)
FSMC_DECLARE_KT_DUAL(48)
FSMC_DECLARE_KT_DUAL(64)
FSMC_DECLARE_KT_DUAL(69)
FSMC_DECLARE_KT_DUAL(80)
FSMC_DECLARE_KT_DUAL(96)
FSMC_DECLARE_KT_DUAL(112)
FSMC_DECLARE_KT_DUAL(128)
#define FSMC_DECLARE_W2_MEMBER(KH, NW) FSMC_W2_KERNELS(FSMC_DECLARE_W2, KH, NW)
FSMC_ALL_W2(FSMC_DECLARE_W2_MEMBER)
#endif

} // namespace fsmc
