// fsmc_inst.hip -- one family member's kernel instantiations (see fsmc_instances.h).
// Compiled with -DFSMC_INSTANCE_KT=<n> (lane-per-pair member) or -DFSMC_INSTANCE_W2=<n> (states per wave of the
// wave-group kernel; -DFSMC_INSTANCE_NW=<waves per group>, four if not given; -DFSMC_INSTANCE_SEQ=0|1: only the array-mode /
// sequence-mode kernels of the member); fastsmc_amd/build.py drives one hipcc per unit, in parallel.
#include "fsmc_instances.h"

namespace fsmc
{
#if defined(FSMC_INSTANCE_KT)
FSMC_KT_KERNELS(FSMC_DEFINE_KT, FSMC_INSTANCE_KT)
FSMC_KT_BIDIR_KERNELS(FSMC_DEFINE_KT_BIDIR, FSMC_INSTANCE_KT)
#if FSMC_INSTANCE_KT > 0
FSMC_DEFINE_KT_DUAL(FSMC_INSTANCE_KT)
#endif
// (every member of the library is built with beta stride 2 as well: halfBuilt(), fsmc_instances.h)
static_assert(halfBuilt(FSMC_INSTANCE_KT), "not a member of the library: add it to FSMC_ALL_KT or FSMC_EXACT_KT");
FSMC_KT_HALF_KERNELS(FSMC_DEFINE_KT, FSMC_INSTANCE_KT)
#if FSMC_INSTANCE_KT != 50 // (halfSumsBuilt, fsmc_instances.h)
FSMC_KT_HALF_SUMS_KERNELS(FSMC_DEFINE_KT, FSMC_INSTANCE_KT)
#endif
FSMC_DEFINE_KT_DUAL_HALF(FSMC_INSTANCE_KT)
#elif defined(FSMC_INSTANCE_W2)
#ifndef FSMC_INSTANCE_NW
#define FSMC_INSTANCE_NW 4
#endif
#if !defined(FSMC_INSTANCE_SEQ)
FSMC_W2_KERNELS(FSMC_DEFINE_W2, FSMC_INSTANCE_W2, FSMC_INSTANCE_NW)
#elif FSMC_INSTANCE_SEQ
FSMC_W2_MODE_KERNELS(FSMC_DEFINE_W2, FSMC_INSTANCE_W2, FSMC_INSTANCE_NW, true)
#else
FSMC_W2_MODE_KERNELS(FSMC_DEFINE_W2, FSMC_INSTANCE_W2, FSMC_INSTANCE_NW, false)
#endif
#else
#error "define FSMC_INSTANCE_KT or FSMC_INSTANCE_W2"
#endif
} // namespace fsmc
