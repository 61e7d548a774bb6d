// fsmc_capi.hip -- host side of libfastsmc_hip.so: the C ABI of include/fastsmc_hip.h.
//
// Owns device memory (model tables, packed haplotypes, work list, per-wave workspace, record
// buffer), validates every shape on the host before a launch (an out-of-bounds kernel can take
// the whole node down), picks the chunking of the beta stream, launches the decode kernel and
// orders the results the way the reference writes them.  No CPU fallback exists: if HIP is not
// usable every entry point fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "fsmc_identify.h"
#include "fsmc_instances.h"

namespace fsmc
{
hipError_t idSortCandidates(hipStream_t stream, const fsmc_candidate* in, fsmc_candidate* sorted, unsigned n,
                            unsigned nHaps, unsigned nWords); // fsmc_identify_sort.hip
hipError_t idSeedDepths(hipStream_t stream, const unsigned long long* words, unsigned nHaps, unsigned nWords,
                        unsigned maxSeeds, unsigned readAhead, unsigned char* depthOut,
                        unsigned hapStride); // fsmc_identify_seeds.hip
}

using namespace fsmc;

namespace fsmc
{
// acc[i] = (((acc[i] + plane_0[i]) + plane_1[i]) + ...): the per-batch sums of one launch are added to the running
// accumulator one batch after the other, the order of sumOverPairs(pos, k) += sum in HMM::augmentSumOverPairs
// (HMM.cpp:1054-1073) -- the same fp32 additions in the same order, hence the same bits.
__global__ void add_planes_in_order_kernel(const float* __restrict__ planes, float* __restrict__ acc, size_t n,
                                           int nPlanes, size_t planeStride)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float s = acc[i];
    for (int sl = 0; sl < nPlanes; ++sl) {
      s = s + planes[(size_t)sl * planeStride + i];
    }
    acc[i] = s;
  }
}
} // namespace fsmc

namespace
{
thread_local std::string g_createError;

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};
} // namespace

struct fsmc_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool ownStream = false;
  int nCU = 0;
  uint64_t hbmBytes = 0;
  uint64_t wsLimit = 0;
  double wsEarned = 0; // bytes of workspace the launches so far (and the announced job) have paid for and not yet
                       // spent on an allocation (earnWorkspace, fsmc_ctx_expect_work, planLaunch)
  double wsAnnounced = 0; // estimated kernel seconds of announced work not launched yet (it has earned already)
  double wsAnnouncedCredit = 0; // ... and the bytes of credit that part of the announcement was given
  std::string err;

  unsigned long long* dHaps = nullptr;
  uint32_t nHaps = 0, nSites = 0, W = 0;

  fsmc_pair* dPairs = nullptr;
  fsmc_group* dGroups = nullptr;
  size_t nPairs = 0, nGroups = 0;
  std::vector<fsmc_pair> hPairs;
  std::vector<fsmc_group> hGroups;

  unsigned* dCounters = nullptr;
  unsigned long long* dPhase = nullptr;
  DevBuf ws;
  DevBuf recs;
  size_t recCap = 0;
  DevBuf aux;  // dump offsets
  DevBuf out;  // device-side result staging (dump / per-pair / sums)

  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timed = false;
  int lastSlots = 0;
  int lastChunk = 0, lastMaxChunks = 0, lastResident = 0;
  int residentChunks = -1; // chunks of a window that skip the rebuild: -1 = as many as the workspace limit allows, 0 = none
  uint32_t chunkSites = 0; // 0 = automatic
  uint32_t betaStride = 0; // 0 = automatic (2 where the kernel exists), 1 = store every beta row
  int lastStride = 1;
  bool lastSpsLds = false; // the last IBD launch kept the segments' per-state sums in LDS
  int lastMember = 0; // member of the last launch: KT of the lane-per-pair kernel; wave-group kernel: 1000 + KH (four waves of KH states), 1000 * NW + KH (NW = 5 ... 8 waves)

  // The queues an IBD decode's waves pull from (fsmc_decode_ibd_launch): built from the uploaded groups once per
  // work list and budget, longest window first.
  struct IbdQueues {
    uint64_t serial = 0;  // work list they were built from
    uint64_t maxLen = 0;  // pairing budgets they were built with: beside a second kernel ...
    uint64_t maxLenAlone = 0; // ... and with the whole workspace (they move with the model member, the beta stride and
                              // the workspace limit, not always together)
    uint32_t pairing = 0;
    bool valid = false;
    bool dual = false;    // items/unions in use: two half-groups per wave
    bool reordered = false; // rest is in use (a reordered copy or subset of the uploaded groups)
    bool alone = false;     // every group rides in the paired kernel: it has the whole workspace
    std::vector<fsmc_group> items, unions, rest;
  } q;
  uint64_t worklistSerial = 0;
  fsmc_group* dItems = nullptr; // two half-groups per wave: {A, B} per item
  fsmc_group* dRest = nullptr;  // the groups that run one per wave, longest window first
  uint32_t pairing = 1;         // 0 = never pair half-full groups, 1 = automatic
  int lastItems = 0;            // items of the last IBD launch (0: it ran on the groups as uploaded)
  unsigned twoWaves = 0;        // two waves per window (fsmc_kernels_bidir.h): 0 automatic, 1 never
  int lastWavesPerWindow = 1;   // ... of the last dump / per-pair / sums launch
  hipStream_t side = nullptr;   // the one-group-per-wave kernel of a paired decode runs here, beside the paired kernel
  hipEvent_t evFork = nullptr, evJoin = nullptr;
  DevBuf wsSide;

  DevBuf idStash;       // fsmc_identify on overflow: the complete, ordered candidate list, kept for fsmc_identify_fetch
  size_t idStashCount = 0;

  const fsmc_model* ibdModel = nullptr;
  uint32_t ibdFlags = 0;
  bool ibdPending = false;
};

struct fsmc_model {
  fsmc_ctx* ctx = nullptr;
  int K = 0, KP = 0, S = 0, nRows = 0;
  int w2NW = 0; // wave-group kernel: waves per group (each holds KP / w2NW states); 0 for every other model
  float *pi = nullptr, *cR = nullptr, *expT = nullptr;
  float *D = nullptr, *B = nullptr, *U = nullptr, *RR = nullptr;
  float* rowSets = nullptr; // [rows][5][KP]: D | B | U | Ush | RR per key, Ush[k] = U[k-1] (kernels' RowSet)
  float* ghostMask = nullptr; // [KP]: 1.0f for the K real states, 0.0f for the padding states
  int* stepRow = nullptr;
  bool sequence = false;
  int *rowGapF = nullptr, *rowSiteB = nullptr, *rowGapB = nullptr; // sequence mode (stepRow = forward site step)
  float4* emis3 = nullptr; // [S][3 or 4][KP]: emission per observation class (+ the gap row in sequence mode)
  unsigned stateThr = 0, ageThr = 0;
  float probThr = 0.f;
};

namespace
{

// Two copies of the HIP runtime in one process (e.g. this library linked against /opt/rocm and a PyTorch wheel that
// bundles its own libamdhip64, loaded in that order) share the device but not their state: kernels built for one
// wave per SIMD then fail to launch with an opaque "unknown error" from the occupancy query.  When that query fails,
// fsmc_ctx_create looks at the mapped files and says so.  Returns the paths of the distinct runtimes found.
std::vector<std::string> mappedHipRuntimes()
{
  std::vector<std::string> found;
  if (FILE* f = std::fopen("/proc/self/maps", "r")) {
    char line[4096];
    while (std::fgets(line, sizeof(line), f)) {
      const char* slash = std::strchr(line, '/');
      if (!slash || !std::strstr(slash, "libamdhip64.so")) {
        continue;
      }
      std::string path(slash);
      while (!path.empty() && (path.back() == '\n' || path.back() == ' ')) {
        path.pop_back();
      }
      if (std::find(found.begin(), found.end(), path) == found.end()) {
        found.push_back(path);
      }
    }
    std::fclose(f);
  }
  return found;
}

int fail(fsmc_ctx* ctx, int code, const std::string& msg)
{
  if (ctx) {
    ctx->err = msg;
  } else {
    g_createError = msg;
  }
  return code;
}

#define FSMC_HIP(ctx, call)                                                                                            \
  do {                                                                                                                 \
    hipError_t e_ = (call);                                                                                            \
    if (e_ != hipSuccess) {                                                                                            \
      return fail((ctx), FSMC_EHIP, std::string(#call) + ": " + hipGetErrorString(e_));                                \
    }                                                                                                                  \
  } while (0)

int ensure(fsmc_ctx* ctx, DevBuf& b, size_t bytes)
{
  if (b.bytes >= bytes && b.p) {
    return FSMC_OK;
  }
  if (b.p) {
    (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
  }
  if (bytes == 0) {
    bytes = 16;
  }
  const bool timing = bytes >= (256u << 20) && std::getenv("FSMC_HOST_TIMING") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  hipError_t e = hipMalloc(&b.p, bytes);
  if (e != hipSuccess) {
    b.p = nullptr;
    return fail(ctx, FSMC_ENOMEM, "hipMalloc of " + std::to_string(bytes) + " bytes failed: " + hipGetErrorString(e));
  }
  if (timing) { // (FSMC_HOST_TIMING: where a job's wall time goes -- allocations of 256 MiB and more)
    std::fprintf(stderr, "[fsmc] hipMalloc of %.2f GB took %.3f s\n", (double)bytes / 1e9,
                 std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  }
  b.bytes = bytes;
  return FSMC_OK;
}

template <typename T> int upload(fsmc_ctx* ctx, T** dst, const T* src, size_t n)
{
  if (*dst) {
    (void)hipFree(*dst);
    *dst = nullptr;
  }
  hipError_t e = hipMalloc((void**)dst, std::max<size_t>(n, 1) * sizeof(T));
  if (e != hipSuccess) {
    *dst = nullptr;
    return fail(ctx, FSMC_ENOMEM, std::string("hipMalloc failed: ") + hipGetErrorString(e));
  }
  if (n) {
    FSMC_HIP(ctx, hipMemcpyAsync(*dst, src, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    FSMC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  return FSMC_OK;
}

// rows padded to KP floats so that every table row starts 16-byte aligned
std::vector<float> padRows(const float* src, size_t rows, int K, int KP)
{
  std::vector<float> out(rows * (size_t)KP, 0.f);
  for (size_t r = 0; r < rows; ++r) {
    std::memcpy(&out[r * KP], src + r * K, sizeof(float) * (size_t)K);
  }
  return out;
}

using KernelFn = void (*)(const KParams);

// Which member of the lane-per-pair family decodes a model (fsmc_instances.h): 69 for the reference's 69-state
// models and K itself where the library has an exact member of K states, the padded row length for every other model
// of at most 128 states, 0 beyond.
int familyMember(const fsmc_model* m)
{
  if (exactMember(m->K)) {
    return m->K;
  }
  return (m->K <= 128 && m->KP % kKPad == 0 && m->KP <= 128) ? m->KP : 0;
}

template <int KT> KernelFn pickMember(int mode, bool track, bool seq, bool half, bool dual = false)
{
  if constexpr (KT > 0 && halfBuilt(KT)) {
    if (dual && half && mode == kModeIbd && !seq) {
      return track ? decode_kernel<KT, kModeIbd, true, false, true, true>
                   : decode_kernel<KT, kModeIbd, false, false, true, true>;
    }
  }
  if constexpr (KT > 0) {
    if (dual && mode == kModeIbd && !seq) {
      return track ? decode_kernel<KT, kModeIbd, true, false, false, true>
                   : decode_kernel<KT, kModeIbd, false, false, false, true>;
    }
  }
  switch (mode) {
  case kModeIbd:
    if constexpr (halfBuilt(KT)) {
      if (half && !seq) {
        return track ? decode_kernel<KT, kModeIbd, true, false, true> : decode_kernel<KT, kModeIbd, false, false, true>;
      }
    }
    if (seq) {
      return track ? decode_kernel<KT, kModeIbd, true, true, false> : decode_kernel<KT, kModeIbd, false, true, false>;
    }
    return track ? decode_kernel<KT, kModeIbd, true, false, false> : decode_kernel<KT, kModeIbd, false, false, false>;
  case kModeDump:
    return seq ? decode_kernel<KT, kModeDump, false, true, false> : decode_kernel<KT, kModeDump, false, false, false>;
  case kModePerPair:
    return seq ? decode_kernel<KT, kModePerPair, false, true, false>
               : decode_kernel<KT, kModePerPair, false, false, false>;
  case kModeSums:
    if constexpr (halfSumsBuilt(KT)) {
      if (half && !seq) {
        return decode_kernel<KT, kModeSums, false, false, true>;
      }
    }
    return seq ? decode_kernel<KT, kModeSums, false, true, false> : decode_kernel<KT, kModeSums, false, false, false>;
  default:
    return nullptr;
  }
}

// Beta stride 2 (every second beta row stored, the others recomputed in the alpha sweep): array-mode IBD decode and
// array-mode sums over pairs (round 5: at size the sums moved 8K bytes a pair-site at the rate the CUs' path to memory
// delivers -- the IBD decode's situation before stride 2) of the family members it is built for.
bool halfAvailable(int mode, const fsmc_model* m)
{
  const int member = familyMember(m);
  return !m->sequence && ((mode == kModeIbd && halfBuilt(member)) || (mode == kModeSums && halfSumsBuilt(member)));
}

// The wide-model kernel with lane = pair and several waves per group (fsmc_kernels_w2.h): 128 < K <= 1024, every
// consumer, array and sequence mode.  fsmc_model_create picks the member (w2Member) and pads such a model's rows to
// KP = waves x states per wave: four waves of 48 or 64 (two workgroups per CU) or 80 states, six to eight waves of 64,
// eight waves of 80 / 96 / 128 beyond 512 states.
bool waveGroups(int mode, const fsmc_model* m)
{
  return m->w2NW > 0 && (mode == kModeIbd || mode == kModeDump || mode == kModeSums || mode == kModePerPair);
}

// Member of the wave-group kernel for a model of K states: {waves per group, states per wave}.  Up to 320 states four
// waves (48, 64: two workgroups per CU; 80: one, with the whole register file); beyond, six to eight waves of 64 states;
// beyond 512 states eight waves of 80 / 96 / 128 states, which have no landing zones for the beta row (they would not
// fit the CU's LDS) and hold part of their vectors in scratch memory (round 5: K = 600 0.03 -> 0.28 of the roofline).
// (Three / four waves of 128 states for 321 ... 512 states were measured too: 1.5 x slower than six / eight of 64,
// profiles/r05_w2_wide_members_of_128.txt.)
// (Measured on the 600 x 3000 list, profiles/r04_wide_members_ab.txt: five waves of 64 are 6 % slower than four of 80
// at 300 / 320 states; four waves of 96 / 112 states -- round 4's first members for 321 ... 448 states, which keep part
// of their vectors in scratch memory -- 10 - 13 % slower than six / seven waves of 64.)
struct W2Member {
  int NW, KH;
};
W2Member w2Member(int K)
{
  // (diagnostic: FSMC_DIAG_W2_MEMBER="<waves>x<states per wave>" picks another member that holds the model -- A/B runs)
  if (const char* v = std::getenv("FSMC_DIAG_W2_MEMBER")) {
    int nw = 0, kh = 0;
    // (a member whose waves the model fills all but the last of, or the 48-state one: where the kernel masks ghosts)
    if (std::sscanf(v, "%dx%d", &nw, &kh) == 2 && nw * kh >= K &&
        ((nw - 1) * kh < K || (kh == 48 && 2 * kh < K) || (nw * kh > 512 && (nw - 2) * kh < K))) {
#define FSMC_IS_W2(KHX, NWX)                                                                                            \
  if (kh == KHX && nw == NWX) {                                                                                        \
    return {nw, kh};                                                                                                   \
  }
      FSMC_ALL_W2(FSMC_IS_W2)
#undef FSMC_IS_W2
    }
  }
  if (K <= 192) return {4, 48};
  if (K <= 256) return {4, 64};
  if (K <= 320) return {4, 80};
  if (K <= 384) return {6, 64};
  if (K <= 448) return {7, 64};
  if (K <= 512) return {8, 64};
  // beyond 512 states: eight waves without landing zones (fsmc_kernels_w2.h, LAND)
  if (K <= 640) return {8, 80};
  if (K <= 768) return {8, 96};
  return {8, 128};
}

template <int KH, int NW, bool SEQ> KernelFn pickWaveGroupKernelOf(int mode, bool track)
{
  if (mode == kModeIbd) {
    return track ? decode_kernel_w2<KH, kModeIbd, true, SEQ, NW> : decode_kernel_w2<KH, kModeIbd, false, SEQ, NW>;
  }
  if (mode == kModeSums) {
    return decode_kernel_w2<KH, kModeSums, false, SEQ, NW>;
  }
  if (mode == kModePerPair) {
    return decode_kernel_w2<KH, kModePerPair, false, SEQ, NW>;
  }
  return decode_kernel_w2<KH, kModeDump, false, SEQ, NW>;
}
template <int KH, int NW> KernelFn pickWaveGroupKernel(int mode, bool track, bool seq)
{
  return seq ? pickWaveGroupKernelOf<KH, NW, true>(mode, track) : pickWaveGroupKernelOf<KH, NW, false>(mode, track);
}

// threads of a workgroup of the kernel pickKernel returns for this mode and model
unsigned blockThreads(int mode, const fsmc_model* m)
{
  return waveGroups(mode, m) ? (unsigned)(m->w2NW * kWave) : (unsigned)kWave;
}

// more than 1024 states: the any-K kernel (fsmc_kernels_any.h)
bool anyStates(const fsmc_model* m)
{
  return m->K > kMaxStatesW2;
}

template <bool SEQ> KernelFn pickAnyKernel(int mode, bool track)
{
  switch (mode) {
  case kModeIbd:
    return track ? decode_kernel_any<kModeIbd, true, SEQ> : decode_kernel_any<kModeIbd, false, SEQ>;
  case kModeDump:
    return decode_kernel_any<kModeDump, false, SEQ>;
  case kModePerPair:
    return decode_kernel_any<kModePerPair, false, SEQ>;
  case kModeSums:
    return decode_kernel_any<kModeSums, false, SEQ>;
  default:
    return nullptr;
  }
}

KernelFn pickKernel(int mode, bool track, const fsmc_model* m, bool dual = false)
{
  if (anyStates(m)) {
    if (mode == kModeIbd) {
      m->ctx->lastStride = 1;
    }
    m->ctx->lastMember = 0;
    return m->sequence ? pickAnyKernel<true>(mode, track) : pickAnyKernel<false>(mode, track);
  }
  if (waveGroups(mode, m)) {
    if (mode == kModeIbd) {
      m->ctx->lastStride = 1;
    }
    const int NW = m->w2NW, KH = m->KP / NW;
    // 1048 ... 1112: four waves per group of 48 ... 112 states; 5064 ... 8064: five ... eight waves of 64
    m->ctx->lastMember = NW == kW2NW ? 1000 + KH : 1000 * NW + KH; // (2128: two waves of 128)
#define FSMC_PICK_W2(KHX, NWX)                                                                                          \
  if (KH == KHX && NW == NWX) {                                                                                        \
    return pickWaveGroupKernel<KHX, NWX>(mode, track, m->sequence != 0);                                               \
  }
    FSMC_ALL_W2(FSMC_PICK_W2)
#undef FSMC_PICK_W2
    return nullptr;
  }
  const bool half = halfAvailable(mode, m) && m->ctx->betaStride != 1; // (also with two half-groups per wave)
  if (mode == kModeIbd || mode == kModeSums) {
    m->ctx->lastStride = half ? 2 : 1;
  }
  const int member = familyMember(m);
  m->ctx->lastMember = member;
  switch (member) {
#define FSMC_PICK_CASE(KTX)                                                                                             \
  case KTX:                                                                                                            \
    return pickMember<KTX>(mode, track, m->sequence, half, dual);
    FSMC_ALL_KT(FSMC_PICK_CASE)
    FSMC_EXACT_KT(FSMC_PICK_CASE)
#undef FSMC_PICK_CASE
  default:
    return nullptr; // (no such model passes fsmc_model_create: K <= 128 has a member, 128 < K <= 256 the wave-group kernel)
  }
}

// Two waves per window (fsmc_kernels_bidir.h): the dump / per-pair / sums consumers of the lane-per-pair family in
// array mode.  nullptr where the kernel is not built (wave-group and any-K models, sequence mode, the IBD decode).
template <int KT> KernelFn pickBidirMember(int mode)
{
  switch (mode) {
  case kModeDump:
    return decode_kernel_bidir<KT, kModeDump>;
  case kModePerPair:
    return decode_kernel_bidir<KT, kModePerPair>;
  case kModeSums:
    return decode_kernel_bidir<KT, kModeSums>;
  default:
    return nullptr;
  }
}
KernelFn pickBidirKernel(int mode, const fsmc_model* m)
{
  if (m->sequence || anyStates(m) || waveGroups(mode, m)) {
    return nullptr;
  }
  switch (familyMember(m)) {
#define FSMC_PICK_BIDIR(KTX)                                                                                            \
  case KTX:                                                                                                            \
    return pickBidirMember<KTX>(mode);
    FSMC_ALL_KT(FSMC_PICK_BIDIR)
    FSMC_EXACT_KT(FSMC_PICK_BIDIR)
#undef FSMC_PICK_BIDIR
  default:
    return nullptr;
  }
}

struct LaunchPlan {
  int chunk = 0;
  int chunkRows = 0; // rows of the chunk buffer (chunk, or half of it with beta stride 2)
  int maxChunks = 0;
  int residentChunks = 0; // chunk buffers beyond the first: chunks whose rows pass B keeps (no rebuild)
  size_t wsSlot = 0; // float4 per slot
  int slots = 0;
};

// Decide chunking of the beta stream and the number of resident waves.
// `items`: the list the waves will pull from when it is not the uploaded group list.  `paired`: two half-groups per
// wave, `items` holding the union of each pair's windows.
// `share`: the launch runs beside another one and may take 1/share of the workspace limit, in `ws`.
// What a decode may spend on its workspace.
// The most it can have (`hard`): the caller's limit if one is set (fsmc_ctx_set_workspace_limit); otherwise 80 % of the
// card where that much is free (the card has 288 GB; the model, the haplotypes and the records are small), at least 40 %.
// What its plan may be UPGRADED with (`soft`) -- a window kept whole instead of chunked, resident chunks, longer windows
// in the paired kernel: the same when the caller set a limit (the caller knows the job).  Otherwise memory has to be
// earned: hipMalloc costs ~40 ms per GB on this driver (230 GB: 4.4-9 s; tools/malloc_cost.py), a rebuilt chunk costs a
// few per cent of a launch, so a run of a few seconds must not start by allocating the card.  Every launch adds what
// it is expected to save -- kEarnFraction of its estimated kernel time -- at the allocation rate; the buffer is kept
// between launches, so a long job reaches the full card and a short one stays small.
// A bigger buffer is a NEW allocation of its whole size (free + hipMalloc), so growth is amortised and every byte is
// paid for once: an upgrade is considered only when the credit covers twice the buffer held (or the whole hard
// budget), and an allocation is debited from the credit.  Creeping up a resident chunk per flush -- 16 allocations of
// 27 ... 230 GB, 80 s of hipMalloc on a 390-s job -- is what this replaces.  A caller that knows its job announces it
// (fsmc_ctx_expect_work: HMM::decodeAll knows its pair count, HMM.cpp:310-321): the credit of the whole job is there
// at the first launch, and a job long enough to pay for the card allocates it once, at the start.
// `cur`: the buffer about to be re-used (what it holds is paid for).
constexpr double kAllocBytesPerSecond = 25e9; // measured: hipMalloc of 64 / 128 GB takes 2.4 / 5.2 s
constexpr double kEarnFraction = 0.06;        // what resident chunks save of a chunked launch (C2: 1974 -> 1854 ms)
constexpr uint64_t kFreeWorkspace = 24ull << 30; // never argued about (at most a second; only what a plan uses is allocated)

struct WsBudget {
  uint64_t hard, soft;
};

uint64_t freeWorkspace()
{
  if (const char* v = std::getenv("FSMC_DIAG_WS_FREE")) { // tests: make the policy visible on a small problem
    return std::strtoull(v, nullptr, 10);
  }
  return kFreeWorkspace;
}

WsBudget workspaceBudget(const fsmc_ctx* ctx, const DevBuf& cur)
{
  size_t freeB = 0, totalB = 0;
  const bool known = hipMemGetInfo(&freeB, &totalB) == hipSuccess;
  const uint64_t reachable = (uint64_t)freeB + cur.bytes;
  const uint64_t margin = (uint64_t)(0.04 * (double)ctx->hbmBytes);
  if (ctx->wsLimit) {
    // (a limit above what the card has free right now -- another process on it -- is cut to that, if it is at least
    //  a quarter of the limit: below, the caller is told by the allocation failing)
    uint64_t lim = ctx->wsLimit;
    if (known && reachable > margin && reachable - margin < lim && reachable - margin >= lim / 4) {
      lim = reachable - margin;
    }
    return {lim, lim};
  }
  const uint64_t floor40 = (uint64_t)(0.40 * (double)ctx->hbmBytes);
  uint64_t hard = floor40;
  if (known) {
    const uint64_t want = (uint64_t)(0.80 * (double)ctx->hbmBytes);
    hard = std::max<uint64_t>(floor40, std::min<uint64_t>(want, reachable > margin ? reachable - margin : 0));
  }
  uint64_t earned = (uint64_t)std::min(std::max(ctx->wsEarned, 0.0), 1e15);
  if (earned < hard && earned < 2 * (uint64_t)cur.bytes) {
    earned = 0; // (not yet worth a re-allocation: growth is geometric)
  }
  const uint64_t freeBytes = freeWorkspace();
  const uint64_t soft = std::min<uint64_t>(hard, std::max<uint64_t>({(uint64_t)cur.bytes, earned, freeBytes}));
  return {hard, soft};
}

// An allocation of the workspace under the earned policy is debited from the credit (beyond the free allowance).
void payForWorkspace(fsmc_ctx* ctx, uint64_t allocatedBytes)
{
  if (ctx->wsLimit) {
    return;
  }
  const uint64_t freeBytes = freeWorkspace();
  if (allocatedBytes > freeBytes) {
    ctx->wsEarned = std::max(0.0, ctx->wsEarned - (double)(allocatedBytes - freeBytes));
  }
}

double earnScale()
{
  if (const char* v = std::getenv("FSMC_DIAG_WS_EARN_SCALE")) { // tests: a small problem that earns like a long job
    return std::atof(v);
  }
  return 1.0;
}

// A launch over the uploaded work list earns workspace for this and the following launches: estimated kernel time =
// algorithmic bytes at 80 % of the roofline.
void earnWorkspace(fsmc_ctx* ctx, const fsmc_model* m, int mode)
{
  double pairSites = 0;
  for (const fsmc_group& g : ctx->hGroups) {
    const uint32_t aEnd = (mode == kModeIbd) ? g.scan_to : g.to;
    pairSites += (double)g.n_pairs * (double)(aEnd > g.from ? aEnd - g.from : 0);
  }
  double seconds = pairSites * (8.0 * m->K + 0.25) / (0.8 * 8e12);
  // (work that was announced has earned already: fsmc_ctx_expect_work)
  const double announced = std::min(ctx->wsAnnounced, seconds);
  if (ctx->wsAnnounced > 0) {
    ctx->wsAnnouncedCredit *= (ctx->wsAnnounced - announced) / ctx->wsAnnounced; // (that part is spent: it was launched)
  }
  ctx->wsAnnounced -= announced;
  seconds -= announced;
  ctx->wsEarned += earnScale() * kEarnFraction * seconds * kAllocBytesPerSecond;
}

int planLaunch(fsmc_ctx* ctx, const fsmc_model* m, int mode, KernelFn fn, LaunchPlan& plan,
               const std::vector<fsmc_group>* items = nullptr, bool paired = false, unsigned share = 1,
               DevBuf* ws = nullptr)
{
  const std::vector<fsmc_group>& list = items ? *items : ctx->hGroups;
  if (!fn) {
    return fail(ctx, FSMC_EUNSUPPORTED, "no kernel for this model");
  }
  int blocksPerCU = 0;
  FSMC_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocksPerCU, fn, (int)blockThreads(mode, m), 0));
  if (blocksPerCU < 1) {
    blocksPerCU = 1;
  }
  if (blocksPerCU > 8) {
    blocksPerCU = 8;
  }
  if (anyStates(m) && blocksPerCU > 4) {
    // the any-K kernel streams its K-vectors through the caches: with four waves per CU more of them stay there
    // (600 x 3000 list, K = 300: 1 / 2 / 4 / 8 waves per CU run 10.3 / 6.9 / 5.0 / 6.5 s)
    blocksPerCU = 4;
  }
  if (const char* cap = std::getenv("FSMC_DIAG_WAVES_PER_CU")) { // occupancy experiments only
    const int v = std::atoi(cap);
    if (v >= 1 && v < blocksPerCU) {
      blocksPerCU = v;
    }
  }
  const bool w2 = waveGroups(mode, m); // a workgroup of four waves takes a group, lane = pair
  size_t slots = (size_t)ctx->nCU * blocksPerCU;
  slots = std::min(slots, list.size());
  if (slots < 1) {
    slots = 1;
  }
  size_t L = 1;
  for (const fsmc_group& g : list) {
    const size_t aEnd = (mode == kModeIbd) ? g.scan_to : g.to;
    L = std::max<size_t>(L, aEnd - g.from);
  }
  // float4 per lane of a stored K-vector: a padded family member (and the wave-group kernel) stores its ghosts too
  const int member = familyMember(m);
  // (the any-K kernel's rows carry their scale in one more float4)
  const size_t K4 = w2 ? (size_t)m->KP / 4 : (size_t)((member > 0 ? member : m->K) + 3) / 4 + (anyStates(m) ? 1 : 0);
  const size_t vecBytes = K4 * kWave * sizeof(float4);
  const WsBudget budget = workspaceBudget(ctx, ws ? *ws : ctx->ws);
  const uint64_t limit = budget.hard / share;
  // Rows a chunk of C sites needs in the chunk buffer: with beta stride 2 only every second site's row is stored.
  const bool half = halfAvailable(mode, m) && ctx->betaStride != 1;
  auto chunkRows = [&](size_t c) { return half ? (c + 1) / 2 : c; };
  const size_t rowsAvail = (size_t)(limit / (vecBytes * slots)); // rows one resident wave may hold
  const size_t rowsSoft = (size_t)(budget.soft / share / (vecBytes * slots)); // ... and may have earned so far
  // rows beside the chunk buffer and the checkpoints: parking / segment sums; the any-K kernel keeps every vector there
  const size_t sideRows = anyStates(m) ? (size_t)kAnyExtraRows + 2 : 4;
  const size_t defaultChunk = std::max<size_t>((size_t)std::ceil(std::sqrt((double)L)), w2 ? 2048 : 512);
  size_t C, maxChunks;
  // A window stays whole if that fits what the launches have earned (or is no longer than a chunk would be anyway).
  // An explicit chunk length (fsmc_ctx_set_chunk_sites) is kept -- by the paired kernel too, which has the chunked
  // layout since round 3.
  if (chunkRows(L) + sideRows + 1 <= rowsAvail && (chunkRows(L) + sideRows + 1 <= rowsSoft || L <= defaultChunk) &&
      !(ctx->chunkSites && ctx->chunkSites < L)) {
    C = L;
    maxChunks = 1;
  } else {
    // The chunk length of a window that does not fit the workspace whole: 512 sites, or sqrt(window) if that is larger
    // (memory is chunk + window / chunk rows), shrunk to what the workspace limit allows.  Short chunks cost a
    // checkpoint row and a pipeline restart each and let the resident chunks (below) fill the memory that is left to
    // the last row: C2 at 2048 / 1024 / 512 sites a chunk runs 1886 / 1881 / 1870 ms.
    C = ctx->chunkSites ? (size_t)ctx->chunkSites : defaultChunk;
    // (the wave-group kernel pays more per restart: 2048)
    C = std::min((C + 15) / 16 * 16, (L + 15) / 16 * 16);
    auto fits = [&](size_t c) { return chunkRows(c) + (L + c - 1) / c + sideRows + 1 <= rowsAvail; };
    if (!ctx->chunkSites) {
      while (C > 16 && !fits(C)) {
        C = std::max<size_t>(16, (C / 2 + 15) / 16 * 16);
      }
    }
    maxChunks = (L + C - 1) / C;
    if (!fits(C)) {
      return fail(ctx, FSMC_ENOMEM, "workspace limit too small for the decode window");
    }
  }
  // Resident chunks (fsmc_kernels.h): a chunked window rebuilds every chunk's rows from a checkpoint -- one of its 3.5
  // sweeps -- except for the chunks whose rows pass B can leave in the workspace.  Whatever the limit leaves after the
  // one chunk buffer, the checkpoints and the parking rows goes to such chunks (array-mode IBD decode and sums of the
  // lane-per-pair family, one group per wave -- the paired kernel is not built with them -- and of the wave-group kernel's
  // four-wave members).
  size_t resident = 0;
  if (maxChunks > 1 && (mode == kModeIbd || mode == kModeSums) && !m->sequence && !paired && !anyStates(m) &&
      (!w2 || w2ResidentBuilt(m->w2NW, false)) &&
      ctx->residentChunks != 0) {
    const size_t rowsBudget = rowsSoft;
    // (exactly the rows of plan.wsSlot below: a plan must qualify again for the buffer it was given -- with a row of
    //  slack here a context whose soft budget is the buffer it holds lost one resident chunk at its next launch)
    const size_t fixed = chunkRows(C) + maxChunks + sideRows;
    if (rowsBudget > fixed) {
      resident = std::min<size_t>((rowsBudget - fixed) / chunkRows(C), maxChunks);
      if (ctx->residentChunks > 0) {
        resident = std::min<size_t>(resident, (size_t)ctx->residentChunks);
      }
    }
  }
  plan.chunk = (int)C;
  plan.chunkRows = (int)chunkRows(C);
  plan.maxChunks = (int)maxChunks;
  plan.residentChunks = (int)resident;
  plan.wsSlot = (chunkRows(C) * (1 + resident) + maxChunks + 2 + (sideRows - 2)) * K4 * kWave;
  plan.slots = (int)slots;
  ctx->lastChunk = plan.chunk;
  ctx->lastMaxChunks = plan.maxChunks;
  ctx->lastResident = plan.residentChunks;
  DevBuf& buf = ws ? *ws : ctx->ws;
  const size_t held = buf.bytes;
  const int rc = ensure(ctx, buf, plan.wsSlot * sizeof(float4) * slots);
  if (rc == FSMC_OK && buf.bytes != held) {
    payForWorkspace(ctx, buf.bytes);
  }
  return rc;
}

// A launch of the two-waves-per-window kernel, if this one qualifies: `nItems` workgroups (groups; batches of the sums)
// that are ALL resident at once -- so the launch has at most half as many items as the chip holds waves of this member,
// the case in which the one-wave kernel leaves every wave alone on a SIMD (or SIMDs idle) -- and whose whole windows fit
// the workspace (to - from rows a workgroup; the same rule as planLaunch's whole windows: inside the hard budget, and
// inside what the context has earned unless the window is no longer than a chunk would be).
bool planTwoWaves(fsmc_ctx* ctx, const fsmc_model* m, int mode, size_t nItems, KernelFn& fn, LaunchPlan& plan)
{
  ctx->lastWavesPerWindow = 1;
  if (ctx->twoWaves == 1 || nItems == 0) {
    return false;
  }
  if (const char* v = std::getenv("FSMC_DIAG_TWO_WAVE_WINDOWS")) { // tests: the whole suite on the one-wave kernels
    if (std::strcmp(v, "never") == 0) {
      return false;
    }
  }
  KernelFn f = pickBidirKernel(mode, m);
  if (!f) {
    return false;
  }
  int blocksPerCU = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocksPerCU, f, 2 * kWave, 0) != hipSuccess || blocksPerCU < 1) {
    (void)hipGetLastError();
    return false;
  }
  blocksPerCU = std::min(blocksPerCU, 4);
  if (const char* cap = std::getenv("FSMC_DIAG_WAVES_PER_CU")) { // occupancy experiments only
    const int v = std::atoi(cap) / 2;
    if (v >= 1 && v < blocksPerCU) {
      blocksPerCU = v;
    }
  }
  if (nItems > (size_t)ctx->nCU * blocksPerCU) {
    return false;
  }
  size_t L = 1;
  for (const fsmc_group& g : ctx->hGroups) {
    L = std::max<size_t>(L, g.to - g.from);
  }
  const int member = familyMember(m);
  const size_t K4 = (size_t)(member + 3) / 4;
  const size_t vecBytes = K4 * kWave * sizeof(float4);
  const WsBudget budget = workspaceBudget(ctx, ctx->ws);
  const uint64_t need = (uint64_t)L * vecBytes * nItems;
  if (need > budget.hard || (need > budget.soft && L > 512)) {
    return false;
  }
  plan = LaunchPlan();
  plan.chunk = (int)L;
  plan.chunkRows = (int)L;
  plan.maxChunks = 1;
  plan.wsSlot = L * K4 * kWave;
  plan.slots = (int)nItems;
  const size_t held = ctx->ws.bytes;
  if (ensure(ctx, ctx->ws, need) != FSMC_OK) {
    ctx->err.clear();
    return false;
  }
  if (ctx->ws.bytes != held) {
    payForWorkspace(ctx, ctx->ws.bytes);
  }
  ctx->lastChunk = plan.chunk;
  ctx->lastMaxChunks = 1;
  ctx->lastResident = 0;
  ctx->lastWavesPerWindow = 2;
  fn = f;
  return true;
}

int checkReady(fsmc_ctx* ctx, const fsmc_model* m)
{
  if (!ctx || !m) {
    return fail(ctx, FSMC_EINVAL, "null context or model");
  }
  if (m->ctx != ctx) {
    return fail(ctx, FSMC_EINVAL, "model belongs to another context");
  }
  if (!ctx->dHaps) {
    return fail(ctx, FSMC_ESTATE, "no haplotypes uploaded (fsmc_haps_upload)");
  }
  if (!ctx->dPairs || !ctx->dGroups || ctx->nGroups == 0) {
    return fail(ctx, FSMC_ESTATE, "no work list uploaded (fsmc_worklist_upload)");
  }
  if ((uint32_t)m->S != ctx->nSites) {
    return fail(ctx, FSMC_EINVAL, "model has " + std::to_string(m->S) + " sites but haplotypes have " +
                                      std::to_string(ctx->nSites));
  }
  for (const fsmc_pair& pr : ctx->hPairs) {
    if (pr.hap_a >= ctx->nHaps || pr.hap_b >= ctx->nHaps) {
      return fail(ctx, FSMC_EINVAL, "pair refers to a haplotype row outside the uploaded matrix");
    }
  }
  for (const fsmc_group& g : ctx->hGroups) {
    if (g.to > (uint32_t)m->S) {
      return fail(ctx, FSMC_EINVAL, "group window exceeds the number of sites");
    }
  }
  if (m->K > kMaxStatesAny) {
    return fail(ctx, FSMC_EUNSUPPORTED, "more than " + std::to_string(kMaxStatesAny) + " states");
  }
  return FSMC_OK;
}

void fillParams(const fsmc_ctx* ctx, const fsmc_model* m, const LaunchPlan& plan, uint32_t flags, KParams& p)
{
  std::memset(&p, 0, sizeof(p));
  p.K = m->K;
  p.KP = m->KP;
  p.S = m->S;
  p.W = (int)ctx->W;
  p.nGroups = (int)ctx->nGroups;
  p.chunk = plan.chunk;
  p.chunkRows = plan.chunkRows;
  p.maxChunks = plan.maxChunks;
  p.residentChunks = plan.residentChunks;
  p.flags = flags;
  p.pi = m->pi;
  p.cR = m->cR;
  p.expT = m->expT;
  p.D = m->D;
  p.B = m->B;
  p.U = m->U;
  p.rowSets = m->rowSets;
  p.RR = m->RR;
  p.ghostMask = m->ghostMask;
  p.stepRow = m->stepRow;
  p.rowGapF = m->rowGapF;
  p.rowSiteB = m->rowSiteB;
  p.rowGapB = m->rowGapB;
  p.emis3 = m->emis3;
  p.haps = ctx->dHaps;
  p.pairs = ctx->dPairs;
  p.groups = ctx->dGroups;
  p.counters = ctx->dCounters;
  p.phaseCycles = ctx->dPhase;
  p.ws = (float4*)ctx->ws.p;
  p.wsSlot = plan.wsSlot;
  p.stateThr = m->stateThr;
  p.ageThr = m->ageThr;
  p.spsLds = 0u;
  // int * float, evaluated in fp32 like HMM.cpp:1226,1254,1281,1308
  p.thr[0] = 1000 * m->probThr;
  p.thr[1] = 100 * m->probThr;
  p.thr[2] = 10 * m->probThr;
  p.thr[3] = m->probThr;
  p.recs = (fsmc_ibd_record*)ctx->recs.p;
  p.recCap = (unsigned)ctx->recCap;
}

// `continues`: a later launch of one call's sequence (the sums' batches, `slots` a launch): the timed span started with
// the first one and ends behind the last (fsmc_last_kernel_ms: the whole sequence, its plane additions included).
int launch(fsmc_ctx* ctx, KernelFn fn, const KParams& p, int slots, unsigned threads = kWave, size_t dynLds = 0,
           bool continues = false)
{
  FSMC_HIP(ctx, hipMemsetAsync(ctx->dCounters, 0, 4 * sizeof(unsigned), ctx->stream));
  if (!continues) {
    FSMC_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  }
  hipLaunchKernelGGL(fn, dim3((unsigned)slots), dim3(threads), dynLds, ctx->stream, p);
  FSMC_HIP(ctx, hipGetLastError());
  FSMC_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  ctx->timed = true;
  ctx->lastSlots = slots;
  return FSMC_OK;
}

// One decode as two kernels side by side: `pSide` (one group per wave; the long windows, so it starts first) on the
// side stream and `pMain` (two half-groups per wave) on the context's stream.  They share the record buffer and its
// counter and have a queue head each; the timed span covers both.
int launchBeside(fsmc_ctx* ctx, KernelFn fnSide, KParams& pSide, int slotsSide, KernelFn fnMain, KParams& pMain,
                 int slotsMain)
{
  FSMC_HIP(ctx, hipMemsetAsync(ctx->dCounters, 0, 4 * sizeof(unsigned), ctx->stream));
  FSMC_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  FSMC_HIP(ctx, hipEventRecord(ctx->evFork, ctx->stream));
  FSMC_HIP(ctx, hipStreamWaitEvent(ctx->side, ctx->evFork, 0));
  pMain.groupBase = 0;
  pSide.groupBase = 2;
  hipLaunchKernelGGL(fnSide, dim3((unsigned)slotsSide), dim3(kWave), 0, ctx->side, pSide);
  FSMC_HIP(ctx, hipGetLastError());
  hipLaunchKernelGGL(fnMain, dim3((unsigned)slotsMain), dim3(kWave), 0, ctx->stream, pMain);
  FSMC_HIP(ctx, hipGetLastError());
  FSMC_HIP(ctx, hipEventRecord(ctx->evJoin, ctx->side));
  FSMC_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->evJoin, 0));
  FSMC_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  ctx->timed = true;
  ctx->lastSlots = slotsSide + slotsMain;
  return FSMC_OK;
}

} // namespace

extern "C" {

const char* fsmc_last_error(const fsmc_ctx* ctx)
{
  return ctx ? ctx->err.c_str() : g_createError.c_str();
}

int fsmc_ctx_create(int device_id, void* stream, fsmc_ctx** out)
{
  if (!out) {
    return fail(nullptr, FSMC_EINVAL, "out is null");
  }
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    return fail(nullptr, FSMC_ENODEVICE,
                std::string("no HIP device available (") + (e == hipSuccess ? "count = 0" : hipGetErrorString(e)) +
                    "); this library has no CPU fallback");
  }
  if (device_id < 0 || device_id >= n) {
    return fail(nullptr, FSMC_EINVAL, "device_id out of range");
  }
  fsmc_ctx* ctx = new (std::nothrow) fsmc_ctx();
  if (!ctx) {
    return fail(nullptr, FSMC_ENOMEM, "host allocation failed");
  }
  ctx->device = device_id;
  e = hipSetDevice(device_id);
  hipDeviceProp_t prop;
  if (e == hipSuccess) {
    e = hipGetDeviceProperties(&prop, device_id);
  }
  if (e != hipSuccess) {
    delete ctx;
    return fail(nullptr, FSMC_ENODEVICE, std::string("hipSetDevice/hipGetDeviceProperties: ") + hipGetErrorString(e));
  }
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    std::string arch = prop.gcnArchName;
    delete ctx;
    return fail(nullptr, FSMC_ENODEVICE, "device is " + arch + "; this library is built for gfx950 (MI355X) only");
  }
  ctx->nCU = prop.multiProcessorCount;
  ctx->hbmBytes = (uint64_t)prop.totalGlobalMem;
  if (stream) {
    ctx->stream = (hipStream_t)stream;
  } else {
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    ctx->ownStream = (e == hipSuccess);
  }
  if (e == hipSuccess) e = hipEventCreate(&ctx->ev0);
  if (e == hipSuccess) e = hipEventCreate(&ctx->ev1);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->evFork, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->evJoin, hipEventDisableTiming);
  if (e == hipSuccess) e = hipMalloc((void**)&ctx->dCounters, 4 * sizeof(unsigned));
  if (e == hipSuccess) e = hipMalloc((void**)&ctx->dPhase, kPhaseSlots * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMemset(ctx->dPhase, 0, kPhaseSlots * sizeof(unsigned long long));
  if (e != hipSuccess) {
    std::string msg = std::string("context set-up failed: ") + hipGetErrorString(e);
    fsmc_ctx_destroy(ctx);
    return fail(nullptr, FSMC_EHIP, msg);
  }
  // Two copies of the HIP runtime in the process (mappedHipRuntimes) are a DIAGNOSIS, not a gate: an embedding process
  // may map a second copy and still launch fine.  So try what breaks in that case -- the occupancy query of a kernel
  // built for one wave per SIMD -- and name the cause only if it does.
  {
    int blocks = 0;
    const hipError_t pe = hipOccupancyMaxActiveBlocksPerMultiprocessor(
      &blocks, decode_kernel<128, kModeIbd, true, false, false>, kWave, 0);
    if (pe != hipSuccess) {
      (void)hipGetLastError();
      const std::vector<std::string> rts = mappedHipRuntimes();
      std::string msg = std::string("kernels of this library cannot be launched (") + hipGetErrorString(pe) + ")";
      int code = FSMC_EHIP;
      if (rts.size() > 1) {
        code = FSMC_ERUNTIME;
        msg += ": two HIP runtimes are loaded in this process (";
        for (size_t i = 0; i < rts.size(); ++i) {
          msg += (i ? ", " : "") + rts[i];
        }
        msg += "): load one libamdhip64 only -- e.g. import torch before this library so that both resolve to the "
               "copy torch bundles, or link both against the same ROCm";
      }
      fsmc_ctx_destroy(ctx);
      return fail(nullptr, code, msg);
    }
  }
  *out = ctx;
  return FSMC_OK;
}

void fsmc_ctx_destroy(fsmc_ctx* ctx)
{
  if (!ctx) {
    return;
  }
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) {
    (void)hipStreamSynchronize(ctx->stream);
  }
  if (ctx->dHaps) (void)hipFree(ctx->dHaps);
  if (ctx->dPairs) (void)hipFree(ctx->dPairs);
  if (ctx->dGroups) (void)hipFree(ctx->dGroups);
  if (ctx->dItems) (void)hipFree(ctx->dItems);
  if (ctx->dRest) (void)hipFree(ctx->dRest);
  if (ctx->dCounters) (void)hipFree(ctx->dCounters);
  if (ctx->dPhase) (void)hipFree(ctx->dPhase);
  if (ctx->idStash.p) (void)hipFree(ctx->idStash.p);
  if (ctx->ws.p) (void)hipFree(ctx->ws.p);
  if (ctx->recs.p) (void)hipFree(ctx->recs.p);
  if (ctx->aux.p) (void)hipFree(ctx->aux.p);
  if (ctx->out.p) (void)hipFree(ctx->out.p);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->evFork) (void)hipEventDestroy(ctx->evFork);
  if (ctx->evJoin) (void)hipEventDestroy(ctx->evJoin);
  if (ctx->side) (void)hipStreamDestroy(ctx->side);
  if (ctx->wsSide.p) (void)hipFree(ctx->wsSide.p);
  if (ctx->ownStream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

int fsmc_ctx_info(const fsmc_ctx* ctx, int32_t* n_cu, int32_t* n_slots, uint64_t* hbm_bytes)
{
  if (!ctx) {
    return FSMC_EINVAL;
  }
  if (n_cu) *n_cu = ctx->nCU;
  if (n_slots) *n_slots = ctx->lastSlots;
  if (hbm_bytes) *hbm_bytes = ctx->hbmBytes;
  return FSMC_OK;
}

int fsmc_ctx_set_workspace_limit(fsmc_ctx* ctx, uint64_t bytes)
{
  if (!ctx) {
    return FSMC_EINVAL;
  }
  ctx->wsLimit = bytes;
  return FSMC_OK;
}

int fsmc_ctx_expect_work(fsmc_ctx* ctx, double pair_sites, int32_t states)
{
  if (!ctx) {
    return fail(nullptr, FSMC_EINVAL, "null context");
  }
  if (!(pair_sites >= 0) || states < 1) {
    return fail(ctx, FSMC_EINVAL, "expected work: pair-sites >= 0 and states >= 1");
  }
  if (pair_sites == 0) {
    // the announced job is over (HMM::finishDecoding / finishFromHashing): what is left of the announcement -- an
    // estimate that was too high, pairs that were filtered, a job that threw -- is forgotten and its unspent credit
    // taken back, so that a later job on this context earns from its own launches again
    ctx->wsEarned = std::max(0.0, ctx->wsEarned - ctx->wsAnnouncedCredit);
    ctx->wsAnnounced = 0;
    ctx->wsAnnouncedCredit = 0;
    return FSMC_OK;
  }
  // the job's launches will save what workspace upgrades save of its estimated kernel time: its credit is there at the
  // first launch (workspaceBudget); the launches themselves then earn nothing until the announced work is used up.
  // No announcement is worth more than the card: the credit is capped at the device's memory.
  const double seconds = pair_sites * (8.0 * states + 0.25) / (0.8 * 8e12);
  const double credit = std::min(earnScale() * kEarnFraction * seconds * kAllocBytesPerSecond,
                                 std::max(0.0, (double)ctx->hbmBytes - ctx->wsAnnouncedCredit));
  ctx->wsAnnounced += seconds;
  ctx->wsAnnouncedCredit += credit;
  ctx->wsEarned += credit;
  return FSMC_OK;
}

int fsmc_ctx_set_chunk_sites(fsmc_ctx* ctx, uint32_t sites)
{
  if (!ctx) {
    return FSMC_EINVAL;
  }
  ctx->chunkSites = sites;
  return FSMC_OK;
}

int fsmc_ctx_set_beta_stride(fsmc_ctx* ctx, uint32_t stride)
{
  if (!ctx || stride > 2) {
    return fail(ctx, FSMC_EINVAL, "beta stride must be 0 (automatic), 1 or 2");
  }
  ctx->betaStride = stride;
  return FSMC_OK;
}

int fsmc_ctx_last_beta_stride(const fsmc_ctx* ctx, int32_t* stride)
{
  if (!ctx || !stride) {
    return FSMC_EINVAL;
  }
  *stride = ctx->lastStride;
  return FSMC_OK;
}

int fsmc_ctx_last_plan(const fsmc_ctx* ctx, int32_t* chunk_sites, int32_t* max_chunks, int32_t* n_slots)
{
  if (!ctx) {
    return FSMC_EINVAL;
  }
  if (chunk_sites) *chunk_sites = ctx->lastChunk;
  if (max_chunks) *max_chunks = ctx->lastMaxChunks;
  if (n_slots) *n_slots = ctx->lastSlots;
  return FSMC_OK;
}

int fsmc_ctx_set_resident_chunks(fsmc_ctx* ctx, int32_t chunks)
{
  if (!ctx || chunks < -1) {
    return fail(ctx, FSMC_EINVAL, "resident chunks: -1 (automatic), 0 (none) or a count");
  }
  ctx->residentChunks = chunks;
  return FSMC_OK;
}

int fsmc_ctx_last_resident_chunks(const fsmc_ctx* ctx, int32_t* chunks)
{
  if (!ctx || !chunks) {
    return FSMC_EINVAL;
  }
  *chunks = ctx->lastResident;
  return FSMC_OK;
}

int fsmc_ctx_last_kernel(const fsmc_ctx* ctx, int32_t* member)
{
  if (!ctx || !member) {
    return FSMC_EINVAL;
  }
  *member = ctx->lastMember;
  return FSMC_OK;
}

int fsmc_model_create(fsmc_ctx* ctx, const fsmc_model_desc* d, fsmc_model** out)
{
  if (!ctx || !d || !out) {
    return fail(ctx, FSMC_EINVAL, "null argument");
  }
  *out = nullptr;
  if (d->K < 2 || d->S < 1 || d->n_rows < 1) {
    return fail(ctx, FSMC_EINVAL, "need K >= 2, S >= 1, n_rows >= 1");
  }
  if (d->K > kMaxStatesAny) {
    return fail(ctx, FSMC_EUNSUPPORTED, "more than " + std::to_string(kMaxStatesAny) + " states");
  }
  if (!d->pi || !d->col_ratios || !d->exp_times || !d->D || !d->B || !d->U || !d->RR || !d->step_row || !d->e1 ||
      !d->e0m1 || !d->e2m0) {
    return fail(ctx, FSMC_EINVAL, "null table pointer in model description");
  }
  if (d->state_threshold > (uint32_t)d->K || d->age_threshold > (uint32_t)d->K) {
    return fail(ctx, FSMC_EINVAL, "state/age threshold larger than K");
  }
  const bool seq = d->sequence != 0;
  if (seq && (!d->gap_row_f || !d->site_row_f || !d->gap_row_b || !d->site_row_b || !d->hom)) {
    return fail(ctx, FSMC_EINVAL, "sequence mode needs gap_row_f, site_row_f, gap_row_b, site_row_b and hom");
  }
  const int32_t* rowArrays[5] = {d->step_row, seq ? d->gap_row_f : nullptr, seq ? d->site_row_f : nullptr,
                                 seq ? d->gap_row_b : nullptr, seq ? d->site_row_b : nullptr};
  static const char* const rowNames[5] = {"step_row", "gap_row_f", "site_row_f", "gap_row_b", "site_row_b"};
  for (int a = 0; a < 5; ++a) {
    for (int32_t s = 1; rowArrays[a] && s < d->S; ++s) {
      if (rowArrays[a][s] < 0 || rowArrays[a][s] >= d->n_rows) {
        return fail(ctx, FSMC_EINVAL,
                    std::string(rowNames[a]) + "[" + std::to_string(s) + "] outside the transition tables");
      }
    }
  }
  FSMC_HIP(ctx, hipSetDevice(ctx->device));
  fsmc_model* m = new (std::nothrow) fsmc_model();
  if (!m) {
    return fail(ctx, FSMC_ENOMEM, "host allocation failed");
  }
  m->ctx = ctx;
  m->K = d->K;
  m->KP = (d->K + kKPad - 1) / kKPad * kKPad; // rows zero padded to whole operand blocks of any tunable width
  if (d->K > 128 && d->K <= kMaxStatesW2) {
    // wide models: NW waves per group hold KP / NW states each (fsmc_kernels_w2.h); the padding states are ghosts
    const W2Member w = w2Member(d->K);
    m->w2NW = w.NW;
    m->KP = w.NW * w.KH;
  }
  m->S = d->S;
  m->nRows = d->n_rows;
  m->stateThr = d->state_threshold;
  m->ageThr = d->age_threshold;
  m->probThr = d->probability_threshold;
  const int K = m->K, KP = m->KP;
  int rc = FSMC_OK;
  auto up = [&](float** dst, const float* src, size_t rows) {
    if (rc == FSMC_OK) {
      std::vector<float> padded = padRows(src, rows, K, KP);
      rc = upload(ctx, dst, padded.data(), padded.size());
    }
  };
  up(&m->pi, d->pi, 1);
  up(&m->cR, d->col_ratios, 1);
  up(&m->expT, d->exp_times, 1);
  up(&m->D, d->D, (size_t)d->n_rows);
  up(&m->B, d->B, (size_t)d->n_rows);
  up(&m->U, d->U, (size_t)d->n_rows);
  up(&m->RR, d->RR, (size_t)d->n_rows);
  if (rc == FSMC_OK) {
    // RowSet copy for the packed steps: the rows of one key side by side so that one base register and
    // immediate offsets address all of them.  Ush is U moved up one state: the packed beta step multiplies
    // vec[k] = beta[k]*e[k] by U[k-1] (the term U[k-1]*vec[k] of BU[k-1], HMM.cpp:986-1005).
    const size_t n = (size_t)d->n_rows;
    std::vector<float> rs(n * kRowSetParts * (size_t)KP, 0.f);
    for (size_t r = 0; r < n; ++r) {
      float* q = &rs[r * kRowSetParts * KP];
      for (int k = 0; k < K; ++k) {
        q[kRowD * KP + k] = d->D[r * K + k];
        q[kRowB * KP + k] = d->B[r * K + k];
        q[kRowU * KP + k] = d->U[r * K + k];
        q[kRowRR * KP + k] = d->RR[r * K + k];
        if (k >= 1) {
          q[kRowUsh * KP + k] = d->U[r * K + (k - 1)];
        }
      }
    }
    rc = upload(ctx, &m->rowSets, rs.data(), rs.size());
  }
  if (rc == FSMC_OK) {
    std::vector<float> mask((size_t)KP, 0.f);
    std::fill(mask.begin(), mask.begin() + K, 1.0f);
    rc = upload(ctx, &m->ghostMask, mask.data(), mask.size());
  }
  m->sequence = seq;
  auto upRows = [&](int** dst, const int32_t* src) {
    if (rc == FSMC_OK) {
      std::vector<int> rows(src, src + d->S);
      rows[0] = 0;
      rc = upload(ctx, dst, rows.data(), rows.size());
    }
  };
  // the kernel's "stepRow" is the (forward) site step: step_row in array mode, site_row_f in sequence mode
  upRows(&m->stepRow, seq ? d->site_row_f : d->step_row);
  if (seq) {
    upRows(&m->rowGapF, d->gap_row_f);
    upRows(&m->rowSiteB, d->site_row_b);
    upRows(&m->rowGapB, d->gap_row_b);
  }
  if (rc == FSMC_OK) {
    // The reference evaluates e = (e1 + e0m1*isZero) + e2m0*isTwo with isZero/isTwo in {0,1}
    // (HMM.cpp:827-828, 959-961).  The three reachable (isZero,isTwo) combinations are tabulated
    // here with the same fp32 expression, so the kernel's row select is bit-identical.
    // Sequence mode adds a fourth row per site: the homozygous emission of the gap before it, which the reference
    // passes as all three emission vectors with all-zero observations (HMM.cpp:764-766): (h + h*0) + h*0.
    const int NC = seq ? 4 : 3;
    std::vector<float> emis((size_t)d->S * NC * KP, 0.f);
    static const float zs[3] = {0.f, 1.f, 1.f};
    static const float ts[3] = {0.f, 0.f, 1.f};
    for (int32_t s = 0; s < d->S; ++s) {
      for (int c = 0; c < 3; ++c) {
        float* dst = &emis[((size_t)s * NC + c) * KP];
        const volatile float z = zs[c];
        const volatile float t = ts[c];
        for (int k = 0; k < K; ++k) {
          const size_t i = (size_t)s * K + k;
          dst[k] = d->e1[i] + d->e0m1[i] * z + d->e2m0[i] * t;
        }
      }
      if (seq) {
        float* dst = &emis[((size_t)s * NC + 3) * KP];
        const volatile float zero = 0.f;
        for (int k = 0; k < K; ++k) {
          const float h = d->hom[(size_t)s * K + k];
          dst[k] = h + h * zero + h * zero;
        }
      }
    }
    rc = upload(ctx, (float**)&m->emis3, emis.data(), emis.size());
  }
  if (rc != FSMC_OK) {
    fsmc_model_destroy(m);
    return rc;
  }
  *out = m;
  return FSMC_OK;
}

void fsmc_model_destroy(fsmc_model* m)
{
  if (!m) {
    return;
  }
  if (m->ctx) {
    (void)hipSetDevice(m->ctx->device);
    if (m->ctx->ibdModel == m) {
      m->ctx->ibdModel = nullptr;
    }
  }
  float* ptrs[] = {m->pi, m->cR, m->expT, m->D, m->B, m->U, m->rowSets, m->ghostMask, m->RR, (float*)m->emis3};
  for (float* q : ptrs) {
    if (q) (void)hipFree(q);
  }
  int* rowPtrs[] = {m->stepRow, m->rowGapF, m->rowSiteB, m->rowGapB};
  for (int* q : rowPtrs) {
    if (q) (void)hipFree(q);
  }
  delete m;
}

int fsmc_haps_upload(fsmc_ctx* ctx, const uint64_t* bits, uint32_t n_haps, uint32_t n_sites)
{
  if (!ctx || !bits || n_haps == 0 || n_sites == 0) {
    return fail(ctx, FSMC_EINVAL, "null or empty haplotype matrix");
  }
  FSMC_HIP(ctx, hipSetDevice(ctx->device));
  const uint32_t W = (n_sites + 63u) / 64u;
  int rc = upload(ctx, &ctx->dHaps, (const unsigned long long*)bits, (size_t)n_haps * W);
  if (rc != FSMC_OK) {
    return rc;
  }
  ctx->nHaps = n_haps;
  ctx->nSites = n_sites;
  ctx->W = W;
  return FSMC_OK;
}

int fsmc_worklist_upload(fsmc_ctx* ctx, const fsmc_pair* pairs, size_t n_pairs, const fsmc_group* groups,
                         size_t n_groups)
{
  if (!ctx || !pairs || !groups || n_pairs == 0 || n_groups == 0) {
    return fail(ctx, FSMC_EINVAL, "null or empty work list");
  }
  if (n_pairs > 0xFFFFFFF0ull) {
    return fail(ctx, FSMC_EINVAL, "too many pairs for one work list");
  }
  size_t covered = 0;
  for (size_t g = 0; g < n_groups; ++g) {
    const fsmc_group& G = groups[g];
    if (G.n_pairs < 1 || G.n_pairs > (uint32_t)kWave) {
      return fail(ctx, FSMC_EINVAL, "group " + std::to_string(g) + ": n_pairs must be 1..64");
    }
    if (G.first_pair != covered) {
      return fail(ctx, FSMC_EINVAL, "group " + std::to_string(g) + ": groups must partition the pair list in order");
    }
    if (!(G.from < G.to) || !(G.from <= G.scan_from && G.scan_from < G.scan_to && G.scan_to <= G.to)) {
      return fail(ctx, FSMC_EINVAL, "group " + std::to_string(g) + ": need from <= scan_from < scan_to <= to");
    }
    covered += G.n_pairs;
  }
  if (covered != n_pairs) {
    return fail(ctx, FSMC_EINVAL, "groups cover " + std::to_string(covered) + " pairs, list has " +
                                      std::to_string(n_pairs));
  }
  FSMC_HIP(ctx, hipSetDevice(ctx->device));
  int rc = upload(ctx, &ctx->dPairs, pairs, n_pairs);
  if (rc == FSMC_OK) {
    rc = upload(ctx, &ctx->dGroups, groups, n_groups);
  }
  if (rc != FSMC_OK) {
    return rc;
  }
  ctx->nPairs = n_pairs;
  ctx->nGroups = n_groups;
  ctx->hPairs.assign(pairs, pairs + n_pairs);
  ctx->hGroups.assign(groups, groups + n_groups);
  ctx->worklistSerial += 1;
  ctx->ibdPending = false;
  return FSMC_OK;
}

namespace
{
// Two half-groups per wavefront.  A group of at most 32 pairs (a hashing-mode batch) fills half a wave; two of them can
// share one if the wave walks the union of their windows (decode_kernel<..., DUAL>).  Worth it when the union is not
// much longer than the longer window: groups are sorted by (window length, from) and neighbours are paired while the
// union stays within 1.25 x the longer one.  Returns false when the
// list does not call for it (no half-full groups, or all of them were packed by the caller already).
bool buildDualItems(const std::vector<fsmc_group>& groups, uint32_t maxLen, std::vector<fsmc_group>& items,
                    std::vector<fsmc_group>& unions, std::vector<fsmc_group>& rest)
{
  // half-full groups whose window (and so the union with a neighbour of the same length class) is within `maxLen`
  // are candidates (the paired kernel decodes in chunks too, so the callers pass no real bound any more); everything
  // else -- full groups -- runs as uploaded, in a second kernel
  std::vector<uint32_t> small;
  for (size_t g = 0; g < groups.size(); ++g) {
    if (groups[g].n_pairs <= 32 && (uint64_t)(groups[g].to - groups[g].from) * 5 <= (uint64_t)maxLen * 4) {
      small.push_back((uint32_t)g);
    }
  }
  if (small.size() < 2) {
    return false;
  }
  // neighbours after this sort have windows of (nearly) the same length that start close to each other: the union a
  // pair walks is then little longer than either window (lengths are compared in 64-site steps, the hashing word)
  std::sort(small.begin(), small.end(), [&](uint32_t x, uint32_t y) {
    const fsmc_group &a = groups[x], &b = groups[y];
    const uint32_t la = (a.to - a.from + 63) / 64, lb = (b.to - b.from + 63) / 64;
    return la != lb ? la < lb : a.from != b.from ? a.from < b.from : x < y;
  });
  items.clear();
  unions.clear();
  rest.clear();
  std::vector<char> taken(groups.size(), 0);
  for (size_t i = 0; i + 1 < small.size();) {
    const fsmc_group &a = groups[small[i]], &b = groups[small[i + 1]];
    const uint32_t lenA = a.to - a.from, lenB = b.to - b.from;
    const uint32_t lenU = std::max(a.to, b.to) - std::min(a.from, b.from);
    if ((uint64_t)lenU * 4 <= (uint64_t)std::max(lenA, lenB) * 5 && lenU <= maxLen) {
      items.push_back(a);
      items.push_back(b);
      fsmc_group u = a;
      u.from = std::min(a.from, b.from);
      u.to = std::max(a.to, b.to);
      u.scan_from = std::min(a.scan_from, b.scan_from);
      u.scan_to = std::max(a.scan_to, b.scan_to);
      unions.push_back(u);
      taken[small[i]] = taken[small[i + 1]] = 1;
      i += 2;
    } else {
      i += 1;
    }
  }
  if (unions.empty()) {
    return false;
  }
  // A half-full group that found no partner but fits the layout rides in the same kernel as an item of its own (B
  // empty): one queue, longest window first, instead of a second kernel whose long windows finish whenever the
  // hardware got round to starting them.  What does not fit (more than 32 pairs, a window beyond the budget) is `rest`.
  for (size_t g = 0; g < groups.size(); ++g) {
    if (taken[g]) {
      continue;
    }
    const fsmc_group& a = groups[g];
    if (a.n_pairs <= 32 && a.to - a.from <= maxLen) {
      fsmc_group none = a;
      none.n_pairs = 0;
      items.push_back(a);
      items.push_back(none);
      unions.push_back(a);
    } else {
      rest.push_back(a);
    }
  }
  // the waves pull the items longest first
  std::vector<uint32_t> order(unions.size());
  for (size_t i = 0; i < order.size(); ++i) {
    order[i] = (uint32_t)i;
  }
  std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
    return unions[x].scan_to - unions[x].from > unions[y].scan_to - unions[y].from;
  });
  std::vector<fsmc_group> items2(items.size()), unions2(unions.size());
  for (size_t i = 0; i < order.size(); ++i) {
    unions2[i] = unions[order[i]];
    items2[2 * i] = items[2 * order[i]];
    items2[2 * i + 1] = items[2 * order[i] + 1];
  }
  items.swap(items2);
  unions.swap(unions2);
  return true;
}

// Longest window first: the decode ends when the last wave does, and a wave that pulls a long window late in the
// queue runs on alone.  (The order of the queue is free: records are keyed by pair and sorted at fetch.)
// Returns false when the list already is in that order (every window the same length, as in the all-pairs modes).
bool longestFirst(std::vector<fsmc_group>& list)
{
  auto len = [](const fsmc_group& g) { return g.scan_to - g.from; };
  bool ordered = true;
  for (size_t i = 1; i < list.size() && ordered; ++i) {
    ordered = len(list[i]) <= len(list[i - 1]);
  }
  if (ordered) {
    return false;
  }
  std::stable_sort(list.begin(), list.end(), [&](const fsmc_group& a, const fsmc_group& b) { return len(a) > len(b); });
  return true;
}
} // namespace

int fsmc_ctx_set_pairing(fsmc_ctx* ctx, uint32_t mode)
{
  if (!ctx || mode > 1) {
    return fail(ctx, FSMC_EINVAL, "pairing must be 0 (off) or 1 (automatic)");
  }
  ctx->pairing = mode;
  return FSMC_OK;
}

int fsmc_ctx_set_two_wave_windows(fsmc_ctx* ctx, uint32_t mode)
{
  if (!ctx || mode > 1) {
    return fail(ctx, FSMC_EINVAL, "two-wave windows: 0 (automatic) or 1 (never)");
  }
  ctx->twoWaves = mode;
  return FSMC_OK;
}

int fsmc_ctx_last_waves_per_window(const fsmc_ctx* ctx, int32_t* waves)
{
  if (!ctx || !waves) {
    return FSMC_EINVAL;
  }
  *waves = ctx->lastWavesPerWindow;
  return FSMC_OK;
}

int fsmc_ctx_last_items(const fsmc_ctx* ctx, int32_t* n_items)
{
  if (!ctx || !n_items) {
    return FSMC_EINVAL;
  }
  *n_items = ctx->lastItems;
  return FSMC_OK;
}

int fsmc_ctx_last_segment_sums_in_lds(const fsmc_ctx* ctx, int32_t* in_lds)
{
  if (!ctx || !in_lds) {
    return FSMC_EINVAL;
  }
  *in_lds = ctx->lastSpsLds ? 1 : 0;
  return FSMC_OK;
}

int fsmc_decode_ibd_launch(fsmc_ctx* ctx, const fsmc_model* m, uint32_t flags)
{
  int rc = checkReady(ctx, m);
  if (rc != FSMC_OK) {
    return rc;
  }
  FSMC_HIP(ctx, hipSetDevice(ctx->device));
  const bool track = (flags & (FSMC_WANT_MEAN | FSMC_WANT_MAP)) != 0;
  earnWorkspace(ctx, m, kModeIbd);
  // The queues.  Two half-groups per wave for the half-full groups that pair up
  // (decode_kernel<..., DUAL>, with beta stride 2 where that is built); the other groups one per wave, in a kernel that
  // runs beside it.
  KernelFn fnDual = nullptr;
  uint64_t maxLen = 0, maxLenAlone = 0; // pairing budgets: beside a second kernel (half the workspace) / on its own
  if (ctx->pairing != 0 && !m->sequence && familyMember(m) > 0) {
    fnDual = pickKernel(kModeIbd, track, m, true);
    // (the paired kernel has the chunked layout too: how long a window may be is no longer a question of memory)
    maxLen = maxLenAlone = 1u << 30;
  }
  fsmc_ctx::IbdQueues& q = ctx->q;
  if (!q.valid || q.serial != ctx->worklistSerial || q.maxLen != maxLen || q.maxLenAlone != maxLenAlone ||
      q.pairing != ctx->pairing) {
    q.valid = false;
    q.items.clear();
    q.unions.clear();
    q.rest.clear();
    // first with the whole workspace: if every group then rides in the paired kernel there is no second kernel to share with
    q.dual = maxLenAlone > 0 && buildDualItems(ctx->hGroups, (uint32_t)maxLenAlone, q.items, q.unions, q.rest);
    q.alone = q.dual && q.rest.empty();
    if (q.dual && !q.alone) {
      q.items.clear();
      q.unions.clear();
      q.rest.clear();
      q.dual = maxLen > 0 && buildDualItems(ctx->hGroups, (uint32_t)maxLen, q.items, q.unions, q.rest);
    }
    if (!q.dual) {
      q.items.clear();
      q.unions.clear();
      q.rest = ctx->hGroups;
    }
    q.reordered = longestFirst(q.rest) || q.dual;
    if (q.dual) {
      rc = upload(ctx, &ctx->dItems, q.items.data(), q.items.size());
      if (rc != FSMC_OK) {
        return rc;
      }
    }
    if (q.reordered && !q.rest.empty()) {
      rc = upload(ctx, &ctx->dRest, q.rest.data(), q.rest.size());
      if (rc != FSMC_OK) {
        return rc;
      }
    }
    q.serial = ctx->worklistSerial;
    q.maxLen = maxLen;
    q.maxLenAlone = maxLenAlone;
    q.pairing = ctx->pairing;
    q.valid = true;
  }
  const bool haveRest = !q.rest.empty();
  const bool beside = q.dual && haveRest;
  LaunchPlan planDual, plan;
  if (q.dual) {
    rc = planLaunch(ctx, m, kModeIbd, fnDual, planDual, &q.unions, true, q.alone ? 1 : 2);
    if (rc != FSMC_OK) {
      return rc;
    }
  }
  KernelFn fn = pickKernel(kModeIbd, track, m, false); // (after the paired pick: last_kernel / last_beta_stride name this one)
  if (haveRest) {
    rc = planLaunch(ctx, m, kModeIbd, fn, plan, q.reordered ? &q.rest : nullptr, false, beside ? 2 : 1,
                    beside ? &ctx->wsSide : nullptr);
    if (rc != FSMC_OK) {
      return rc;
    }
  }
  ctx->lastItems = q.dual ? (int)q.unions.size() : 0;
  if (ctx->recCap == 0) {
    ctx->recCap = std::max<size_t>(1u << 16, 8 * ctx->nPairs);
  }
  rc = ensure(ctx, ctx->recs, ctx->recCap * sizeof(fsmc_ibd_record));
  if (rc != FSMC_OK) {
    return rc;
  }
  KParams pDual, p;
  if (q.dual) {
    fillParams(ctx, m, planDual, flags, pDual);
    pDual.groups = ctx->dItems;
    pDual.nGroups = (int)q.unions.size();
  }
  if (haveRest) {
    fillParams(ctx, m, plan, flags, p);
    if (q.reordered) {
      p.groups = ctx->dRest;
      p.nGroups = (int)q.rest.size();
    }
    if (beside) {
      p.ws = (float4*)ctx->wsSide.p;
    }
  }
  // A launch smaller than the chip (one group per wave, segment ages wanted): where every wave of it still finds room, the
  // open segments' per-state sums live in LDS instead of the workspace (fsmc_kernels.h, KParams::spsLds) -- K4 KiB of
  // dynamic LDS a wave.  (A wave alone on its SIMD waits out every round trip of those sums to L2: C1 shape 40.6 -> 3x ms.)
  size_t spsLdsBytes = 0;
  if (!beside && !q.dual && track && !waveGroups(kModeIbd, m) && !anyStates(m) && !m->sequence) {
    const int member = familyMember(m);
    const size_t dyn = (size_t)((member > 0 ? member : m->K) + 3) / 4 * kWave * sizeof(float4);
    hipFuncAttributes fa;
    FSMC_HIP(ctx, hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(fn)));
    const size_t perWave = fa.sharedSizeBytes + dyn;
    constexpr size_t kLdsPerCU = 160u << 10, kLdsPerWorkgroup = 64u << 10;
    if (perWave <= kLdsPerWorkgroup && (size_t)plan.slots * perWave <= (size_t)ctx->nCU * kLdsPerCU &&
        (size_t)plan.slots <= (size_t)ctx->nCU * (kLdsPerCU / perWave) && !std::getenv("FSMC_DIAG_NO_SPS_LDS")) {
      spsLdsBytes = dyn;
      p.spsLds = 1u;
    }
  }
  ctx->lastSpsLds = spsLdsBytes != 0;
  rc = beside ? launchBeside(ctx, fn, p, plan.slots, fnDual, pDual, planDual.slots)
              : q.dual ? launch(ctx, fnDual, pDual, planDual.slots)
                       : launch(ctx, fn, p, plan.slots, blockThreads(kModeIbd, m), spsLdsBytes);
  if (rc != FSMC_OK) {
    return rc;
  }
  ctx->ibdModel = m;
  ctx->ibdFlags = flags;
  ctx->ibdPending = true;
  return FSMC_OK;
}

int fsmc_identify(fsmc_ctx* ctx, const uint64_t* words, uint32_t n_haps, uint32_t n_words, const uint32_t* global_ids,
                  const fsmc_job_window* job, const float* gen_pos, uint32_t n_sites, int32_t gap, float skip,
                  float min_m, fsmc_candidate* out, size_t cap, size_t* n_out)
{
  const fsmc_identify_opts defaults = {64u, 1u, 0, 10u}; // DecodingParams.hpp: hashingWordSize, haploid, max_seeds,
                                                         // constReadAhead
  return fsmc_identify_ex(ctx, words, n_haps, n_words, global_ids, job, gen_pos, n_sites, gap, skip, min_m, &defaults,
                          out, cap, n_out);
}

int fsmc_identify_ex(fsmc_ctx* ctx, const uint64_t* words, uint32_t n_haps, uint32_t n_words,
                     const uint32_t* global_ids, const fsmc_job_window* job, const float* gen_pos, uint32_t n_sites,
                     int32_t gap, float skip, float min_m, const fsmc_identify_opts* opts, fsmc_candidate* out,
                     size_t cap, size_t* n_out)
{
  if (!ctx) {
    return FSMC_EINVAL;
  }
  if (!opts) {
    return fail(ctx, FSMC_EINVAL, "fsmc_identify: opts is null");
  }
  if (opts->word_size < 1 || opts->word_size > 64) {
    return fail(ctx, FSMC_EINVAL, "fsmc_identify: word_size must be 1..64 (a word is one 64-bit integer)");
  }
  const bool splitSeeds = opts->max_seeds > 0; // (SeedHash.hpp:75 compares as unsigned long: a negative one never splits)
  if (splitSeeds && (opts->read_ahead < 1 || opts->read_ahead > 32)) {
    return fail(ctx, FSMC_EINVAL, "fsmc_identify: read_ahead must be 1..32 when max_seeds is set");
  }
  if (!opts->haploid && (n_haps & 1u)) {
    return fail(ctx, FSMC_EINVAL, "fsmc_identify: haploid = 0 pairs rows 2k, 2k+1 -- n_haps must be even");
  }
  if (!n_out) {
    return fail(ctx, FSMC_EINVAL, "fsmc_identify: n_out is null");
  }
  *n_out = 0;
  if (n_haps < 2 || n_words == 0) {
    return FSMC_OK; // no pair, or no complete word (FastSMC.cpp:186-195 hashes complete words only)
  }
  if (!words || !global_ids || !job || !gen_pos || (cap && !out)) {
    return fail(ctx, FSMC_EINVAL, "fsmc_identify: null argument");
  }
  if ((uint64_t)n_words * opts->word_size > n_sites) {
    return fail(ctx, FSMC_EINVAL, "fsmc_identify: n_words * word_size exceeds n_sites");
  }
  if (gap < 0 || cap > 0xFFFFFFFFull) {
    return fail(ctx, FSMC_EINVAL, "fsmc_identify: gap < 0 or cap beyond 2^32");
  }
  FSMC_HIP(ctx, hipSetDevice(ctx->device));
  const unsigned tiles = (n_haps + kIdTile - 1) / kIdTile;
  const unsigned chunks = (n_words + kIdChunk - 1) / kIdChunk;
  // device buffers of this call (one call per job: no caching)
  struct Bufs {
    void* p[8] = {};
    ~Bufs()
    {
      for (void* q : p) {
        if (q) {
          (void)hipFree(q);
        }
      }
    }
  } b;
  const size_t bytes[8] = {(size_t)n_haps * n_words * sizeof(uint64_t),
                           (size_t)n_haps * sizeof(uint32_t),
                           (size_t)n_sites * sizeof(float),
                           (size_t)chunks * tiles * kIdTile * sizeof(unsigned),
                           (size_t)chunks * sizeof(unsigned),
                           std::max<size_t>(cap, 1) * sizeof(fsmc_candidate),
                           sizeof(unsigned),
                           splitSeeds ? (size_t)chunks * kIdChunk * tiles * kIdTile : 16};
  for (int i = 0; i < 8; ++i) {
    const hipError_t e = hipMalloc(&b.p[i], bytes[i]);
    if (e != hipSuccess) {
      b.p[i] = nullptr;
      return fail(ctx, FSMC_ENOMEM, std::string("fsmc_identify: hipMalloc failed: ") + hipGetErrorString(e));
    }
  }
  FSMC_HIP(ctx, hipMemcpyAsync(b.p[0], words, bytes[0], hipMemcpyHostToDevice, ctx->stream));
  FSMC_HIP(ctx, hipMemcpyAsync(b.p[1], global_ids, bytes[1], hipMemcpyHostToDevice, ctx->stream));
  FSMC_HIP(ctx, hipMemcpyAsync(b.p[2], gen_pos, bytes[2], hipMemcpyHostToDevice, ctx->stream));
  FSMC_HIP(ctx, hipMemsetAsync(b.p[3], 0, bytes[3], ctx->stream));
  FSMC_HIP(ctx, hipMemsetAsync(b.p[6], 0, bytes[6], ctx->stream));
  IdParams p;
  p.words = (const unsigned long long*)b.p[0];
  p.nHaps = n_haps;
  p.nWords = n_words;
  p.globalId = (const unsigned*)b.p[1];
  p.job = *job;
  p.gen = (const float*)b.p[2];
  p.nSites = n_sites;
  p.gap = gap;
  p.skip = skip;
  p.minM = min_m;
  p.dupBits = (unsigned*)b.p[3];
  p.hapStride = tiles * kIdTile;
  p.usedBits = (unsigned*)b.p[4];
  p.out = (fsmc_candidate*)b.p[5];
  p.cap = (unsigned)cap;
  p.count = (unsigned*)b.p[6];
  p.wordSize = opts->word_size;
  p.depth = splitSeeds ? (const unsigned char*)b.p[7] : nullptr;
  // id_match_kernel is the default case (haplotype pairs, whole seeds); the other settings take the general kernel
  const bool general = splitSeeds || !opts->haploid;
  const bool haploid = opts->haploid != 0;
  auto launchMatch = [&]() {
    if (!general) {
      hipLaunchKernelGGL(id_match_kernel, dim3(tiles, tiles), dim3(kIdThreads), 0, ctx->stream, p);
    } else if (haploid) {
      hipLaunchKernelGGL(id_match_general_kernel<true>, dim3(tiles, tiles), dim3(kIdThreads), 0, ctx->stream, p);
    } else {
      hipLaunchKernelGGL(id_match_general_kernel<false>, dim3(tiles, tiles), dim3(kIdThreads), 0, ctx->stream, p);
    }
    return hipGetLastError();
  };
  FSMC_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  if (splitSeeds) {
    FSMC_HIP(ctx, hipMemsetAsync(b.p[7], 0, bytes[7], ctx->stream));
    const hipError_t e = idSeedDepths(ctx->stream, p.words, n_haps, n_words, (unsigned)opts->max_seeds, opts->read_ahead,
                                      (unsigned char*)b.p[7], p.hapStride);
    if (e != hipSuccess) {
      return fail(ctx, e == hipErrorOutOfMemory ? FSMC_ENOMEM : FSMC_EHIP,
                  std::string("fsmc_identify: splitting the seeds failed: ") + hipGetErrorString(e));
    }
  }
  hipLaunchKernelGGL(id_dup_kernel, dim3(tiles, tiles), dim3(kIdThreads), 0, ctx->stream, p);
  FSMC_HIP(ctx, hipGetLastError());
  hipLaunchKernelGGL(id_complexity_kernel, dim3(chunks), dim3(kIdThreads), 0, ctx->stream, p);
  FSMC_HIP(ctx, hipGetLastError());
  FSMC_HIP(ctx, launchMatch());
  FSMC_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  ctx->timed = true;
  unsigned count = 0;
  FSMC_HIP(ctx, hipMemcpyAsync(&count, b.p[6], sizeof(count), hipMemcpyDeviceToHost, ctx->stream));
  FSMC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *n_out = count;
  ctx->idStashCount = 0;
  if (count > cap) {
    // The caller's buffer is too small -- the normal case of a first call, nobody knows the count beforehand.  Only the
    // last pass depends on the buffer: run it again into one of the right size (the duplicate and complexity bits are
    // still there), order the records and keep them on the device for fsmc_identify_fetch, so that the caller does not
    // pay the uploads and the two other passes twice.
    void* full = nullptr;
    hipError_t e = hipMalloc(&full, (size_t)count * sizeof(fsmc_candidate));
    if (e == hipSuccess) {
      p.out = (fsmc_candidate*)full;
      p.cap = count;
      e = hipMemsetAsync(b.p[6], 0, bytes[6], ctx->stream);
      if (e == hipSuccess) {
        e = launchMatch();
      }
      if (e == hipSuccess && ensure(ctx, ctx->idStash, (size_t)count * sizeof(fsmc_candidate)) == FSMC_OK) {
        e = idSortCandidates(ctx->stream, (const fsmc_candidate*)full, (fsmc_candidate*)ctx->idStash.p, count, n_haps,
                             n_words);
        if (e == hipSuccess) {
          e = hipStreamSynchronize(ctx->stream);
        }
        if (e == hipSuccess) {
          ctx->idStashCount = count;
        }
      }
      (void)hipFree(full);
    }
    if (ctx->idStashCount == 0) {
      // the second run did not go through (e.g. no memory for the full list): nothing is kept, the caller calls again
      // with a buffer of *n_out records -- and must not find this failure's error state in front of that call's checks
      (void)hipGetLastError();
    }
    return fail(ctx, FSMC_EOVERFLOW, "fsmc_identify: candidate buffer too small");
  }
  if (count) {
    // the emission order: by the word at which the reference's ExtendHash reports a match, then by pair key (the
    // appends of the workgroups arrive in no particular order) -- sorted on the device, fsmc_identify_sort.hip
    void* sorted = nullptr;
    hipError_t e = hipMalloc(&sorted, (size_t)count * sizeof(fsmc_candidate));
    if (e != hipSuccess) {
      return fail(ctx, FSMC_ENOMEM, std::string("fsmc_identify: hipMalloc failed: ") + hipGetErrorString(e));
    }
    e = idSortCandidates(ctx->stream, (const fsmc_candidate*)b.p[5], (fsmc_candidate*)sorted, count, n_haps, n_words);
    if (e == hipSuccess) {
      e = hipMemcpy(out, sorted, (size_t)count * sizeof(fsmc_candidate), hipMemcpyDeviceToHost);
    }
    (void)hipFree(sorted);
    if (e != hipSuccess) {
      return fail(ctx, FSMC_EHIP, std::string("fsmc_identify: ordering the candidates failed: ") + hipGetErrorString(e));
    }
  }
  return FSMC_OK;
}

int fsmc_identify_fetch(fsmc_ctx* ctx, fsmc_candidate* out, size_t cap, size_t* n_out)
{
  if (!ctx || !n_out) {
    return fail(ctx, FSMC_EINVAL, "fsmc_identify_fetch: null argument");
  }
  *n_out = ctx->idStashCount;
  if (ctx->idStashCount == 0) {
    return fail(ctx, FSMC_ESTATE, "fsmc_identify_fetch: no candidate list is kept (it follows an fsmc_identify that "
                                  "returned FSMC_EOVERFLOW)");
  }
  if (cap < ctx->idStashCount || !out) {
    return fail(ctx, FSMC_EOVERFLOW, "fsmc_identify_fetch: candidate buffer too small");
  }
  FSMC_HIP(ctx, hipSetDevice(ctx->device));
  FSMC_HIP(ctx, hipMemcpy(out, ctx->idStash.p, ctx->idStashCount * sizeof(fsmc_candidate), hipMemcpyDeviceToHost));
  ctx->idStashCount = 0;
  (void)hipFree(ctx->idStash.p);
  ctx->idStash = DevBuf{};
  return FSMC_OK;
}

int fsmc_sync(fsmc_ctx* ctx)
{
  if (!ctx) {
    return FSMC_EINVAL;
  }
  FSMC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return FSMC_OK;
}

int fsmc_last_kernel_ms(fsmc_ctx* ctx, float* ms)
{
  if (!ctx || !ms) {
    return FSMC_EINVAL;
  }
  if (!ctx->timed) {
    return fail(ctx, FSMC_ESTATE, "no launch has been timed yet");
  }
  FSMC_HIP(ctx, hipEventSynchronize(ctx->ev1));
  FSMC_HIP(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
  return FSMC_OK;
}

int fsmc_phase_cycles(fsmc_ctx* ctx, uint64_t* out, size_t n)
{
  if (!ctx || !out || n > (size_t)kPhaseSlots) {
    return fail(ctx, FSMC_EINVAL, "bad argument");
  }
  FSMC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  FSMC_HIP(ctx, hipMemcpy(out, ctx->dPhase, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
  FSMC_HIP(ctx, hipMemset(ctx->dPhase, 0, kPhaseSlots * sizeof(unsigned long long)));
  return FSMC_OK;
}

int fsmc_decode_ibd_fetch(fsmc_ctx* ctx, fsmc_ibd_record* out, size_t cap, size_t* n_out)
{
  if (!ctx || !n_out) {
    return fail(ctx, FSMC_EINVAL, "null argument");
  }
  if (!ctx->ibdPending || !ctx->ibdModel) {
    return fail(ctx, FSMC_ESTATE, "no IBD decode in flight");
  }
  FSMC_HIP(ctx, hipSetDevice(ctx->device));
  unsigned counters[4] = {0, 0, 0, 0};
  for (int attempt = 0; attempt < 3; ++attempt) {
    FSMC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    FSMC_HIP(ctx, hipMemcpy(counters, ctx->dCounters, sizeof(counters), hipMemcpyDeviceToHost));
    if (counters[1] <= ctx->recCap) {
      break;
    }
    // the device buffer was too small: grow it and decode again (results are deterministic)
    ctx->recCap = (size_t)counters[1] + (size_t)counters[1] / 8 + 1024;
    int rc = fsmc_decode_ibd_launch(ctx, ctx->ibdModel, ctx->ibdFlags);
    if (rc != FSMC_OK) {
      return rc;
    }
  }
  const size_t n = counters[1];
  if (n > ctx->recCap) {
    return fail(ctx, FSMC_EHIP, "record buffer still too small after regrowing");
  }
  *n_out = n;
  if (n > cap || (n && !out)) {
    return fail(ctx, FSMC_EOVERFLOW, "output buffer holds " + std::to_string(cap) + " records, need " +
                                         std::to_string(n));
  }
  if (n) {
    FSMC_HIP(ctx, hipMemcpy(out, ctx->recs.p, n * sizeof(fsmc_ibd_record), hipMemcpyDeviceToHost));
    // the reference writes batch by batch, pair by pair, site by site (HMM.cpp:1181,1206)
    std::sort(out, out + n, [](const fsmc_ibd_record& x, const fsmc_ibd_record& y) {
      return x.pair != y.pair ? x.pair < y.pair : x.start < y.start;
    });
  }
  return FSMC_OK;
}

int fsmc_decode_ibd(fsmc_ctx* ctx, const fsmc_model* m, const fsmc_pair* pairs, size_t n_pairs,
                    const fsmc_group* groups, size_t n_groups, uint32_t flags, fsmc_ibd_record* out, size_t cap,
                    size_t* n_out)
{
  int rc = fsmc_worklist_upload(ctx, pairs, n_pairs, groups, n_groups);
  if (rc == FSMC_OK) {
    rc = fsmc_decode_ibd_launch(ctx, m, flags);
  }
  if (rc == FSMC_OK) {
    rc = fsmc_decode_ibd_fetch(ctx, out, cap, n_out);
  }
  return rc;
}

int fsmc_decode_posteriors(fsmc_ctx* ctx, const fsmc_model* m, float* out, size_t out_floats)
{
  int rc = checkReady(ctx, m);
  if (rc != FSMC_OK) {
    return rc;
  }
  if (!out) {
    return fail(ctx, FSMC_EINVAL, "out is null");
  }
  FSMC_HIP(ctx, hipSetDevice(ctx->device));
  std::vector<size_t> offsets(ctx->nGroups);
  size_t total = 0;
  for (size_t g = 0; g < ctx->nGroups; ++g) {
    offsets[g] = total;
    total += (size_t)kWave * m->K * (ctx->hGroups[g].to - ctx->hGroups[g].from);
  }
  if (total > out_floats) {
    return fail(ctx, FSMC_EOVERFLOW, "posterior dump needs " + std::to_string(total) + " floats");
  }
  KernelFn fn = pickKernel(kModeDump, false, m);
  LaunchPlan plan;
  earnWorkspace(ctx, m, kModeDump);
  rc = planLaunch(ctx, m, kModeDump, fn, plan);
  unsigned threads = blockThreads(kModeDump, m);
  if (rc == FSMC_OK && planTwoWaves(ctx, m, kModeDump, ctx->nGroups, fn, plan)) {
    threads = 2 * kWave;
  }
  if (rc == FSMC_OK) rc = ensure(ctx, ctx->aux, offsets.size() * sizeof(size_t));
  if (rc == FSMC_OK) rc = ensure(ctx, ctx->out, total * sizeof(float));
  if (rc != FSMC_OK) {
    return rc;
  }
  FSMC_HIP(ctx, hipMemcpyAsync(ctx->aux.p, offsets.data(), offsets.size() * sizeof(size_t), hipMemcpyHostToDevice,
                               ctx->stream));
  KParams p;
  fillParams(ctx, m, plan, 0, p);
  p.dumpOut = (float*)ctx->out.p;
  p.dumpOffsets = (const size_t*)ctx->aux.p;
  rc = launch(ctx, fn, p, plan.slots, threads);
  if (rc != FSMC_OK) {
    return rc;
  }
  FSMC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  FSMC_HIP(ctx, hipMemcpy(out, ctx->out.p, total * sizeof(float), hipMemcpyDeviceToHost));
  return FSMC_OK;
}

int fsmc_decode_per_pair(fsmc_ctx* ctx, const fsmc_model* m, const float* exp_coal_times, float* mean, int32_t* map)
{
  int rc = checkReady(ctx, m);
  if (rc != FSMC_OK) {
    return rc;
  }
  if (!exp_coal_times || (!mean && !map)) {
    return fail(ctx, FSMC_EINVAL, "need expected coalescence times and at least one output");
  }
  FSMC_HIP(ctx, hipSetDevice(ctx->device));
  KernelFn fn = pickKernel(kModePerPair, false, m);
  LaunchPlan plan;
  earnWorkspace(ctx, m, kModePerPair);
  rc = planLaunch(ctx, m, kModePerPair, fn, plan);
  if (rc != FSMC_OK) {
    return rc;
  }
  unsigned threads = blockThreads(kModePerPair, m);
  if (planTwoWaves(ctx, m, kModePerPair, ctx->nGroups, fn, plan)) {
    threads = 2 * kWave;
  }
  const size_t n = ctx->nPairs * (size_t)m->S;
  const size_t coalBytes = (size_t)m->KP * sizeof(float);
  // layout of the staging buffer: [expCoal KP floats][mean n floats][map n ints]
  rc = ensure(ctx, ctx->out, coalBytes + n * (sizeof(float) + sizeof(int32_t)));
  if (rc != FSMC_OK) {
    return rc;
  }
  std::vector<float> coal((size_t)m->KP, 0.f);
  std::memcpy(coal.data(), exp_coal_times, sizeof(float) * (size_t)m->K);
  char* base = (char*)ctx->out.p;
  FSMC_HIP(ctx, hipMemcpyAsync(base, coal.data(), coalBytes, hipMemcpyHostToDevice, ctx->stream));
  FSMC_HIP(ctx, hipMemsetAsync(base + coalBytes, 0, n * (sizeof(float) + sizeof(int32_t)), ctx->stream));
  KParams p;
  fillParams(ctx, m, plan, 0, p);
  p.expCoal = (const float*)base;
  p.ppMean = mean ? (float*)(base + coalBytes) : nullptr;
  p.ppMap = map ? (int*)(base + coalBytes + n * sizeof(float)) : nullptr;
  rc = launch(ctx, fn, p, plan.slots, threads);
  if (rc != FSMC_OK) {
    return rc;
  }
  FSMC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (mean) {
    FSMC_HIP(ctx, hipMemcpy(mean, p.ppMean, n * sizeof(float), hipMemcpyDeviceToHost));
  }
  if (map) {
    FSMC_HIP(ctx, hipMemcpy(map, p.ppMap, n * sizeof(int32_t), hipMemcpyDeviceToHost));
  }
  return FSMC_OK;
}

int fsmc_decode_sums(fsmc_ctx* ctx, const fsmc_model* m, float* sums, float* sums00, float* sums01, float* sums11)
{
  return fsmc_decode_sums_batches(ctx, m, nullptr, 0, sums, sums00, sums01, sums11);
}

int fsmc_decode_sums_batches(fsmc_ctx* ctx, const fsmc_model* m, const uint32_t* batch_first_group, size_t n_batches,
                             float* sums, float* sums00, float* sums01, float* sums11)
{
  int rc = checkReady(ctx, m);
  if (rc != FSMC_OK) {
    return rc;
  }
  if (batch_first_group) {
    // batch b = groups batch_first_group[b] .. batch_first_group[b + 1] - 1: a partition of the group list, in order
    if (n_batches == 0 || batch_first_group[0] != 0 || batch_first_group[n_batches] != ctx->nGroups) {
      return fail(ctx, FSMC_EINVAL, "batches must partition the group list");
    }
    for (size_t bI = 0; bI < n_batches; ++bI) {
      if (batch_first_group[bI + 1] <= batch_first_group[bI]) {
        return fail(ctx, FSMC_EINVAL, "a batch holds at least one group and batches are in group order");
      }
    }
  } else {
    n_batches = ctx->nGroups; // every group is a batch of its own
  }
  const bool mm = sums00 || sums01 || sums11;
  if (!sums && !mm) {
    return fail(ctx, FSMC_EINVAL, "no output requested");
  }
  if (mm && !(sums00 && sums01 && sums11)) {
    return fail(ctx, FSMC_EINVAL, "the 00/01/11 sums come together");
  }
  for (const fsmc_group& g : ctx->hGroups) {
    if (g.from != 0 || g.to != (uint32_t)m->S) {
      return fail(ctx, FSMC_EINVAL, "posterior sums need whole-sequence groups (HMM.cpp:1052)");
    }
  }
  FSMC_HIP(ctx, hipSetDevice(ctx->device));
  KernelFn fn = pickKernel(kModeSums, false, m);
  bool strideForced = false;
  if (fn && ctx->betaStride == 0 && ctx->lastStride == 2) {
    // Beta stride 2 pays where the launch fills the chip (C2: 2762 -> 2454 ms: the rows' traffic was the bound); a wave
    // alone on its SIMD only gets the recomputed half sweep on top (C1 shape: 41.2 -> 43.3 ms).  Left to itself
    // (fsmc_ctx_set_beta_stride 0) a launch of fewer batches than the chip holds waves keeps stride 1 -- the kernel AND
    // its plan (planLaunch reads the same setting).
    int blocksPerCU = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocksPerCU, fn, (int)blockThreads(kModeSums, m), 0) == hipSuccess &&
        blocksPerCU >= 1 && n_batches < (size_t)ctx->nCU * (size_t)std::min(blocksPerCU, 8)) {
      ctx->betaStride = 1;
      strideForced = true;
      fn = pickKernel(kModeSums, false, m);
    }
  }
  LaunchPlan plan;
  earnWorkspace(ctx, m, kModeSums);
  rc = planLaunch(ctx, m, kModeSums, fn, plan);
  if (strideForced) {
    ctx->betaStride = 0;
  }
  if (rc != FSMC_OK) {
    return rc;
  }
  // a launch of few batches (at most half the chip's waves): two waves per window, a workgroup per batch
  const KernelFn fnOne = fn;
  const LaunchPlan planOne = plan;
  unsigned threads = blockThreads(kModeSums, m);
  if (planTwoWaves(ctx, m, kModeSums, n_batches, fn, plan)) {
    threads = 2 * kWave;
  }
  // One launch decodes up to `slots` batches (one per wave; slots = resident waves, fewer if the planes would not fit)
  // and leaves each batch's sums in its own plane (the groups of a batch of more than 64 pairs in turn, each continuing
  // the running sums of the one before); add_planes_in_order_kernel then adds the planes to the accumulator in batch
  // order.  The accumulator starts from the caller's arrays, so that several calls (flushes)
  // continue the same sequential sum.
  const size_t plane = (size_t)m->S * m->K;
  const int nP = mm ? 4 : 1; // planes per group: the sum, or the sum and its 00 / 01 / 11 split
  // what the planes may take: the caller's limit (or a quarter of the card) -- and no more than the card has left
  // beside the workspace this context already holds
  uint64_t limit = ctx->wsLimit ? ctx->wsLimit : (uint64_t)(0.25 * (double)ctx->hbmBytes);
  {
    size_t freeB = 0, totalB = 0;
    if (hipMemGetInfo(&freeB, &totalB) == hipSuccess) {
      const uint64_t room = (uint64_t)freeB + ctx->out.bytes;
      const uint64_t reserve = 2ull << 30;
      limit = std::min<uint64_t>(limit, room > reserve ? room - reserve : 0);
    }
  }
  const size_t slotFloats = (size_t)nP * plane; // a slot holds the planes the launch was asked for
  size_t slots = std::min<size_t>((size_t)plan.slots, n_batches);
  slots = std::max<size_t>(1, std::min<size_t>(slots, limit / (slotFloats * sizeof(float))));
  if (const char* cap = std::getenv("FSMC_DIAG_SUMS_SLOTS")) { // tests: force many launches on a small problem
    const long v = std::atol(cap);
    if (v >= 1 && (size_t)v < slots) {
      slots = (size_t)v;
    }
  }
  if (threads != blockThreads(kModeSums, m) && slots < n_batches) {
    // (the planes of all the batches do not fit one launch: the one-wave kernel, several launches)
    fn = fnOne;
    plan = planOne;
    threads = blockThreads(kModeSums, m);
    ctx->lastWavesPerWindow = 1;
    ctx->lastChunk = plan.chunk;
    ctx->lastMaxChunks = plan.maxChunks;
  }
  rc = ensure(ctx, ctx->out, (slots + 1) * slotFloats * sizeof(float));
  if (rc != FSMC_OK) {
    return rc;
  }
  float* const acc = (float*)ctx->out.p + slots * slotFloats;
  float* dst[4] = {sums, sums00, sums01, sums11};
  for (int q = 0; q < 4; ++q) {
    if (dst[q]) {
      FSMC_HIP(ctx, hipMemcpyAsync(acc + (size_t)q * plane, dst[q], plane * sizeof(float), hipMemcpyHostToDevice,
                                   ctx->stream));
    }
  }
  KParams p;
  fillParams(ctx, m, plan, (sums ? FSMC_WANT_SUMS : 0u) | (mm ? FSMC_WANT_MAJOR_MINOR_SUMS : 0u), p);
  p.sums = (float*)ctx->out.p;
  p.sumsPlane = plane;
  p.sumsSlot = slotFloats;
  if (batch_first_group) {
    rc = ensure(ctx, ctx->aux, (n_batches + 1) * sizeof(uint32_t));
    if (rc != FSMC_OK) {
      return rc;
    }
    FSMC_HIP(ctx, hipMemcpyAsync(ctx->aux.p, batch_first_group, (n_batches + 1) * sizeof(uint32_t), hipMemcpyHostToDevice,
                                 ctx->stream));
    p.batchFirst = (const unsigned*)ctx->aux.p;
  }
  for (size_t base = 0; base < n_batches; base += slots) {
    const size_t n = std::min(slots, n_batches - base);
    p.groupBase = (int)base;
    rc = launch(ctx, fn, p, (int)n, threads, 0, base != 0);
    if (rc != FSMC_OK) {
      return rc;
    }
    // (planes the launch did not ask for hold stale values: they are added to accumulator planes nobody reads)
    hipLaunchKernelGGL(add_planes_in_order_kernel, dim3(1024), dim3(256), 0, ctx->stream, (const float*)ctx->out.p, acc,
                       slotFloats, (int)n, slotFloats);
    FSMC_HIP(ctx, hipGetLastError());
    FSMC_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream)); // (the call's timed span: decode launches + plane additions)
  }
  FSMC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int q = 0; q < 4; ++q) {
    if (dst[q]) {
      FSMC_HIP(ctx, hipMemcpy(dst[q], acc + (size_t)q * plane, plane * sizeof(float), hipMemcpyDeviceToHost));
    }
  }
  return FSMC_OK;
}

} // extern "C"
