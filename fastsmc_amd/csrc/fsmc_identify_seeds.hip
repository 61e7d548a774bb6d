// fsmc_identify_seeds.hip -- max_seeds of the identification step on the device (SeedHash.hpp:41-85).
//
// The reference splits a seed (the haplotypes sharing word c) that has more than max_seeds members by the NEXT word,
// recursively while the words read ahead last, and enumerates the pairs of the sub-seeds only; those pairs are
// extended to the last word looked at.  What a pair needs to know is therefore one small number per (haplotype, word):
// the DEPTH D(h, c) at which h's chain of sub-seeds stops -- the first d with  |{g : words c..c+d of g = those of h}|
// <= max_seeds  or  c + d + 1 >= words read (FastSMC.cpp:186-195: min(n_words, c + read_ahead)).  A pair (a, b) is
// extended at word c iff a and b share the words c .. c + D(a, c) (then D(b, c) = D(a, c)), to word c + D.
//
// The depths come from sorting, one depth at a time over all words at once: per word a segment of n keys, equal keys
// form the seeds.  Depth 0 sorts the word values; depth d sorts (id of the depth d-1 sub-seed, id of the seed of word
// c + d) for the haplotypes still undecided -- two haplotypes share that pair iff they share words c .. c+d.  rocPRIM's
// segmented radix sort; at most read_ahead rounds, each a few milliseconds for 20 000 haplotypes x 800 words.
#include <hip/hip_runtime.h>
#include <rocprim/device/device_segmented_radix_sort.hpp>

namespace fsmc
{

namespace
{

constexpr unsigned char kUndecided = 255;

__global__ void seed_offsets_kernel(unsigned* off, unsigned n, unsigned nWords)
{
  const unsigned c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c <= nWords) {
    off[c] = c * n;
  }
}

// depth 0: key = the word itself
__global__ void seed_keys0_kernel(const unsigned long long* __restrict__ words, unsigned long long* keys,
                                  unsigned* vals, unsigned n, unsigned nWords)
{
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < (size_t)n * nWords) {
    const unsigned c = (unsigned)(idx / n), j = (unsigned)(idx % n);
    keys[idx] = words[(size_t)j * nWords + c];
    vals[idx] = j;
  }
}

// depth d: key = (sub-seed of depth d-1, seed of word c+d) for the undecided; the others become singletons
__global__ void seed_keys_kernel(const unsigned* __restrict__ prev, const unsigned* __restrict__ gid0,
                                 const unsigned char* __restrict__ depth, unsigned long long* keys, unsigned* vals,
                                 unsigned n, unsigned nWords, unsigned d)
{
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < (size_t)n * nWords) {
    const unsigned c = (unsigned)(idx / n), j = (unsigned)(idx % n);
    unsigned long long k = (1ull << 63) | j;
    if (depth[idx] == kUndecided) { // then c + d < words read <= nWords
      k = ((unsigned long long)prev[idx] << 32) | gid0[(size_t)(c + d) * n + j];
    }
    keys[idx] = k;
    vals[idx] = j;
  }
}

// sorted segment -> per haplotype the id (position of the group's first key in the segment) and size of its group
__global__ void seed_groups_kernel(const unsigned long long* __restrict__ keys, const unsigned* __restrict__ vals,
                                   unsigned* gid, unsigned* gsize, unsigned n, unsigned nWords)
{
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)n * nWords) {
    return;
  }
  const unsigned q = (unsigned)(idx % n);
  const unsigned long long k = keys[idx];
  if (q != 0 && keys[idx - 1] == k) {
    return; // not the first of its group
  }
  const size_t segBase = idx - q, segEnd = segBase + n;
  size_t e = idx + 1;
  while (e < segEnd && keys[e] == k) {
    ++e;
  }
  const unsigned size = (unsigned)(e - idx);
  for (size_t m = idx; m < e; ++m) {
    gid[segBase + vals[m]] = q;
    gsize[segBase + vals[m]] = size;
  }
}

// SeedHash.hpp:75: split again while  size > max_seeds && w + 1 < words read,  w = c + d
__global__ void seed_decide_kernel(const unsigned* __restrict__ gsize, unsigned char* depth, unsigned n, unsigned nWords,
                                   unsigned d, unsigned maxSeeds, unsigned readAhead, unsigned* anyUndecided)
{
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < (size_t)n * nWords) {
    if (d != 0 && depth[idx] != kUndecided) {
      return;
    }
    const unsigned c = (unsigned)(idx / n);
    const unsigned long long read = min((unsigned long long)nWords, (unsigned long long)c + readAhead);
    if (gsize[idx] > maxSeeds && (unsigned long long)c + d + 1ull < read) {
      depth[idx] = kUndecided;
      *anyUndecided = 1u; // (same value from every writer)
    } else {
      depth[idx] = (unsigned char)d;
    }
  }
}

// [word][n] -> the padded [word][hapStride] layout the match kernel stages into LDS
__global__ void seed_store_kernel(const unsigned char* __restrict__ depth, unsigned char* out, unsigned n,
                                  unsigned nWords, unsigned hapStride)
{
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < (size_t)n * nWords) {
    const unsigned c = (unsigned)(idx / n), j = (unsigned)(idx % n);
    out[(size_t)c * hapStride + j] = depth[idx];
  }
}

struct Scratch {
  void* p[10] = {};
  ~Scratch()
  {
    for (void* q : p) {
      if (q) {
        (void)hipFree(q);
      }
    }
  }
};

} // namespace

// words: [nHaps][nWords] (device); depthOut: [>= nWords][hapStride] bytes (device, zeroed by the caller).
hipError_t idSeedDepths(hipStream_t stream, const unsigned long long* words, unsigned nHaps, unsigned nWords,
                        unsigned maxSeeds, unsigned readAhead, unsigned char* depthOut, unsigned hapStride)
{
  const size_t total = (size_t)nHaps * nWords;
  if (total == 0 || total > 0xFFFFFFFFull) {
    return hipErrorInvalidValue;
  }
  Scratch s;
  enum { KIN, KOUT, VIN, VOUT, GID0, PREV, GSIZE, DEPTH, OFF, FLAG };
  const size_t bytes[10] = {total * 8, total * 8, total * 4, total * 4, total * 4, total * 4, total * 4, total,
                            ((size_t)nWords + 1) * 4, 4};
  hipError_t e = hipSuccess;
  for (int i = 0; i < 10 && e == hipSuccess; ++i) {
    e = hipMalloc(&s.p[i], bytes[i]);
    if (e != hipSuccess) {
      s.p[i] = nullptr;
    }
  }
  if (e != hipSuccess) {
    return e;
  }
  auto* keysIn = (unsigned long long*)s.p[KIN];
  auto* keysOut = (unsigned long long*)s.p[KOUT];
  auto* valsIn = (unsigned*)s.p[VIN];
  auto* valsOut = (unsigned*)s.p[VOUT];
  auto* gid0 = (unsigned*)s.p[GID0];
  auto* prev = (unsigned*)s.p[PREV];
  auto* gsize = (unsigned*)s.p[GSIZE];
  auto* depth = (unsigned char*)s.p[DEPTH];
  auto* off = (unsigned*)s.p[OFF];
  auto* flag = (unsigned*)s.p[FLAG];
  const unsigned blocks = (unsigned)((total + 255) / 256);
  hipLaunchKernelGGL(seed_offsets_kernel, dim3((nWords + 256) / 256), dim3(256), 0, stream, off, nHaps, nWords);
  size_t tmpBytes = 0;
  e = rocprim::segmented_radix_sort_pairs(nullptr, tmpBytes, keysIn, keysOut, valsIn, valsOut, (unsigned)total, nWords,
                                          off, off + 1, 0, 64, stream);
  void* tmp = nullptr;
  if (e == hipSuccess) {
    e = hipMalloc(&tmp, tmpBytes ? tmpBytes : 16);
  }
  if (e != hipSuccess) {
    return e;
  }
  for (unsigned d = 0; d < readAhead && e == hipSuccess; ++d) {
    if (d == 0) {
      hipLaunchKernelGGL(seed_keys0_kernel, dim3(blocks), dim3(256), 0, stream, words, keysIn, valsIn, nHaps, nWords);
    } else {
      hipLaunchKernelGGL(seed_keys_kernel, dim3(blocks), dim3(256), 0, stream, prev, gid0, depth, keysIn, valsIn, nHaps,
                         nWords, d);
    }
    e = hipGetLastError();
    if (e == hipSuccess) {
      e = rocprim::segmented_radix_sort_pairs(tmp, tmpBytes, keysIn, keysOut, valsIn, valsOut, (unsigned)total, nWords,
                                              off, off + 1, 0, 64, stream);
    }
    if (e != hipSuccess) {
      break;
    }
    hipLaunchKernelGGL(seed_groups_kernel, dim3(blocks), dim3(256), 0, stream, keysOut, valsOut, d == 0 ? gid0 : prev,
                       gsize, nHaps, nWords);
    e = hipMemsetAsync(flag, 0, 4, stream);
    if (e != hipSuccess) {
      break;
    }
    hipLaunchKernelGGL(seed_decide_kernel, dim3(blocks), dim3(256), 0, stream, gsize, depth, nHaps, nWords, d, maxSeeds,
                       readAhead, flag);
    if (d == 0) {
      e = hipMemcpyAsync(prev, gid0, total * 4, hipMemcpyDeviceToDevice, stream);
      if (e != hipSuccess) {
        break;
      }
    }
    unsigned any = 0;
    e = hipMemcpyAsync(&any, flag, 4, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) {
      e = hipStreamSynchronize(stream);
    }
    if (e != hipSuccess || !any) {
      break;
    }
  }
  if (e == hipSuccess) {
    hipLaunchKernelGGL(seed_store_kernel, dim3(blocks), dim3(256), 0, stream, depth, depthOut, nHaps, nWords, hapStride);
    e = hipGetLastError();
    if (e == hipSuccess) {
      e = hipStreamSynchronize(stream);
    }
  }
  (void)hipFree(tmp);
  return e;
}

} // namespace fsmc
