// Micro-benchmark: issue rate of non-packed and packed fp32 VALU ops on gfx950 for 1/2/4 waves per SIMD,
// independent vs dependent chains.  Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE> __global__ void k(float* out, long long* cyc, int iters, float s0, float s1)
{
  float a[8];
  f2 p[8];
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x + i; p[i] = f2{(float)threadIdx.x + i, (float)i}; }
  float d = threadIdx.x;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (MODE == 0) { // 8 independent chains, alternating mul/add (scalar operand)
#pragma unroll
        for (int i = 0; i < 8; ++i) { a[i] = a[i] * s0; a[i] = a[i] + s1; }
      } else if (MODE == 1) { // one dependent chain
#pragma unroll
        for (int i = 0; i < 8; ++i) { d = d * s0; d = d + s1; }
      } else if (MODE == 2) { // packed, 8 independent chains
#pragma unroll
        for (int i = 0; i < 8; ++i) { p[i] = p[i] * f2{s0, s0}; p[i] = p[i] + f2{s1, s1}; }
      }
    }
  }
  long long t1 = clock64();
  float acc = d;
  for (int i = 0; i < 8; ++i) acc += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if ((threadIdx.x & 63) == 0) { cyc[3 * (blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64)] = t1 - t0; cyc[3 * (blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) + 1] = t0; cyc[3 * (blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) + 2] = t1; }
}

template <int MODE> void run(const char* name, int wavesPerSimd, int wavesPerBlock = 1)
{
  const int nCU = 256, iters = 2000;
  const int blocks = nCU * 4 * wavesPerSimd / wavesPerBlock;
  float* out; long long* cyc;
  hipMalloc(&out, blocks * 64 * wavesPerBlock * sizeof(float));
  hipMalloc(&cyc, 3 * blocks * wavesPerBlock * sizeof(long long));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64 * wavesPerBlock), 0, 0, out, cyc, iters, 1.0001f, 0.001f);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64 * wavesPerBlock), 0, 0, out, cyc, iters, 1.0001f, 0.001f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const int nw = blocks * wavesPerBlock;
  long long* h = new long long[3 * nw]; hipMemcpy(h, cyc, 3 * nw * sizeof(long long), hipMemcpyDeviceToHost);
  long long tmin = h[1], tmax = h[2]; double avg = 0;
  for (int i = 0; i < nw; ++i) { if (h[3*i+1] < tmin) tmin = h[3*i+1]; if (h[3*i+2] > tmax) tmax = h[3*i+2]; avg += h[3*i]; }
  avg /= nw;
  const double instrPerWave = (double)iters * 16 * 16; // 16 ops per inner body (8 mul + 8 add)
  printf("%-26s waves/SIMD=%d wpb=%d  per-wave cyc/instr=%.2f  span cyc/instr=%.2f  SIMD instr/cycle=%.3f  (%.3f ms)\n", name,
         wavesPerSimd, wavesPerBlock, avg / instrPerWave, (double)(tmax - tmin) / instrPerWave,
         instrPerWave * wavesPerSimd / (double)(tmax - tmin), ms);
  delete[] h;
  hipFree(out); hipFree(cyc);
}

int main()
{
  for (int w : {1, 2, 4}) run<0>("fp32 mul/add independent", w);
  for (int w : {2, 4}) run<0>("fp32 mul/add independent", w, 4);
  for (int w : {1, 2}) run<1>("fp32 mul/add dependent", w);
  for (int w : {1, 2, 4}) run<2>("packed fp32 independent", w);
  run<2>("packed fp32 independent", 2, 4);
  return 0;
}
