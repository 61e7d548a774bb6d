// fsmc_identify_sort.hip -- the emission order of the identification step's candidates, on the device: records
// sorted by (flush word, lower haplotype * n + higher haplotype) with rocPRIM's radix sort (keys: one 64-bit integer
// per record; values: the record's position), then gathered.  A translation unit of its own: the rocPRIM headers
// take as long to compile as a family member of the decode kernel.
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include "../../include/fastsmc_hip.h"

namespace fsmc
{

__global__ void id_keys_kernel(const fsmc_candidate* in, unsigned long long* keys, unsigned* vals, unsigned n,
                               unsigned long long nHaps)
{
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const fsmc_candidate c = in[i];
    keys[i] = ((unsigned long long)c.flush_word * nHaps + c.hap_a) * nHaps + c.hap_b;
    vals[i] = i;
  }
}

__global__ void id_gather_kernel(const fsmc_candidate* in, const unsigned* vals, fsmc_candidate* out, unsigned n)
{
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    out[i] = in[vals[i]];
  }
}

// in: n records in arrival order; sorted: n records in emission order (both device memory).  n_words bounds flush_word.
hipError_t idSortCandidates(hipStream_t stream, const fsmc_candidate* in, fsmc_candidate* sorted, unsigned n,
                            unsigned nHaps, unsigned nWords)
{
  if (n == 0) {
    return hipSuccess;
  }
  unsigned long long* keys = nullptr;
  unsigned* vals = nullptr;
  void* tmp = nullptr;
  hipError_t e = hipMalloc((void**)&keys, 2 * (size_t)n * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMalloc((void**)&vals, 2 * (size_t)n * sizeof(unsigned));
  size_t tmpBytes = 0;
  // bits of the largest key: (n_words * n + n) * n
  const long double top = ((long double)nWords + 1.0L) * (long double)nHaps * (long double)nHaps;
  unsigned bits = 1;
  while (bits < 64 && (long double)(1ull << bits) <= top) {
    ++bits;
  }
  if (e == hipSuccess) {
    e = rocprim::radix_sort_pairs(nullptr, tmpBytes, keys, keys + n, vals, vals + n, n, 0, bits, stream);
  }
  if (e == hipSuccess) e = hipMalloc(&tmp, tmpBytes ? tmpBytes : 16);
  if (e == hipSuccess) {
    const unsigned blocks = (n + 255) / 256;
    hipLaunchKernelGGL(id_keys_kernel, dim3(blocks), dim3(256), 0, stream, in, keys, vals, n, (unsigned long long)nHaps);
    e = hipGetLastError();
    if (e == hipSuccess) {
      e = rocprim::radix_sort_pairs(tmp, tmpBytes, keys, keys + n, vals, vals + n, n, 0, bits, stream);
    }
    if (e == hipSuccess) {
      hipLaunchKernelGGL(id_gather_kernel, dim3(blocks), dim3(256), 0, stream, in, vals + n, sorted, n);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
  }
  if (tmp) (void)hipFree(tmp);
  if (vals) (void)hipFree(vals);
  if (keys) (void)hipFree(keys);
  return e;
}

} // namespace fsmc
