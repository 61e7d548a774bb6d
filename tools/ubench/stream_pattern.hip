// Micro-benchmark: the decode kernel's HBM access pattern without arithmetic.  Each resident wave owns a private
// chunk buffer; per "site" it writes 18 x 1 KiB rows (non-temporal dwordx4 stores), and in a second phase reads
// them back by LDS-DMA -- chunk by chunk like pass A.  Reports the sustained read+write rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int K4, bool NT> __global__ __launch_bounds__(64, 2) void k(float4* ws, size_t slotF4, int chunk, int nChunks, float* sink)
{
  __shared__ float4 land[K4 * 64];
  float4* buf = ws + (size_t)blockIdx.x * slotF4 + threadIdx.x;
  float acc = 0.f;
  f32x4 v = {1.f + threadIdx.x, 2.f, 3.f, 4.f};
  for (int c = 0; c < nChunks; ++c) {
    for (int s = 0; s < chunk; ++s) {
#pragma unroll
      for (int j = 0; j < K4; ++j) {
        if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(buf + ((size_t)s * K4 + j) * 64));
        else *reinterpret_cast<f32x4*>(buf + ((size_t)s * K4 + j) * 64) = v;
      }
      v.x += 1.f;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    
#if defined(__HIP_DEVICE_COMPILE__)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif

    for (int s = 0; s < chunk; ++s) {
#pragma unroll
      for (int j = 0; j < K4; ++j) {
#if defined(__HIP_DEVICE_COMPILE__)
        if (NT) __builtin_amdgcn_global_load_lds(buf + ((size_t)s * K4 + j) * 64, &land[j * 64], 16, 0, 2);
        else __builtin_amdgcn_global_load_lds(buf + ((size_t)s * K4 + j) * 64, &land[j * 64], 16, 0, 0);
#endif
      }
      
#if defined(__HIP_DEVICE_COMPILE__)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif

      acc += land[(s % K4) * 64 + threadIdx.x].x;
      
#if defined(__HIP_DEVICE_COMPILE__)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif

    }
  }
  sink[blockIdx.x * 64 + threadIdx.x] = acc;
}

template <bool NT> void run(int slots, int chunk, int nChunks)
{
  constexpr int K4 = 18;
  const size_t slotF4 = (size_t)chunk * K4 * 64;
  float4* ws; float* sink;
  hipMalloc(&ws, slotF4 * sizeof(float4) * slots);
  hipMalloc(&sink, slots * 64 * sizeof(float));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<K4, NT>), dim3(slots), dim3(64), 0, 0, ws, slotF4, chunk, 2, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<K4, NT>), dim3(slots), dim3(64), 0, 0, ws, slotF4, chunk, nChunks, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = 2.0 * (double)slots * nChunks * chunk * K4 * 1024.0;
  printf("nt=%d slots=%d chunk=%d: %.3f ms, %.2f TB/s (read+write)\n", (int)NT, slots, chunk, ms, bytes / ms / 1e9);
  hipFree(ws); hipFree(sink);
}

int main()
{
  run<true>(2048, 224, 40);
  run<false>(2048, 224, 40);
  run<true>(1024, 224, 40);
  // small private chunks: does the write -> read-back reuse stay on-die (Infinity Cache)?
  run<false>(2048, 2, 4000);
  run<false>(2048, 4, 2000);
  run<false>(2048, 8, 1000);
  run<false>(2048, 16, 500);
  run<true>(2048, 4, 2000);
  return 0;
}
