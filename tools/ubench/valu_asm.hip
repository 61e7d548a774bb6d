// Micro-benchmark: issue cost of VALU instructions on gfx950 with the instruction mix pinned by inline asm.
//   MODE 0: 8 independent v_mul_f32 (VGPR x SGPR)      MODE 1: one dependent chain of v_mul_f32
//   MODE 2: 8 independent v_pk_mul_f32 (VGPR pair x SGPR pair)   MODE 3: dependent chain of v_pk_mul_f32
//   MODE 4: v_readlane_b32 -> SGPR, 8 independent       MODE 5: two dependent chains interleaved
// Build: hipcc -O3 --offload-arch=gfx950 valu_asm.hip -o valu_asm
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE> __global__ void k(float* out, int iters, float s0)
{
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  double p0 = a0, p1 = a1, p2 = a2, p3 = a3, p4 = a4, p5 = a5, p6 = a6, p7 = a7; // 64-bit register pairs
  double sp = __hip_atomic_load((double*)out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int r0 = 0, r1 = 0, r2 = 0, r3 = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      if (MODE == 0) {
        asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n"
                     "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s0));
      } else if (MODE == 1) {
        asm volatile("v_mul_f32 %0, %1, %0\n v_mul_f32 %0, %1, %0\n v_mul_f32 %0, %1, %0\n v_mul_f32 %0, %1, %0\n"
                     "v_mul_f32 %0, %1, %0\n v_mul_f32 %0, %1, %0\n v_mul_f32 %0, %1, %0\n v_mul_f32 %0, %1, %0"
                     : "+v"(a0) : "s"(s0));
      } else if (MODE == 2) {
        asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                     "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "s"(sp));
      } else if (MODE == 3) {
        asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %0, %0, %1\n"
                     "v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %0, %0, %1"
                     : "+v"(p0) : "s"(sp));
      } else if (MODE == 4) {
        asm volatile("v_readlane_b32 %0, %4, 3\n v_readlane_b32 %1, %4, 5\n v_readlane_b32 %2, %4, 7\n v_readlane_b32 %3, %4, 9\n"
                     "v_readlane_b32 %0, %4, 13\n v_readlane_b32 %1, %4, 15\n v_readlane_b32 %2, %4, 17\n v_readlane_b32 %3, %4, 19"
                     : "+s"(r0), "+s"(r1), "+s"(r2), "+s"(r3) : "v"(a0));
      } else if (MODE == 6) { // packed, both sources VGPR pairs
        asm volatile("v_pk_mul_f32 %0, %0, %7\n v_pk_mul_f32 %1, %1, %7\n v_pk_mul_f32 %2, %2, %7\n v_pk_mul_f32 %3, %3, %7\n"
                     "v_pk_mul_f32 %4, %4, %7\n v_pk_mul_f32 %5, %5, %7\n v_pk_mul_f32 %6, %6, %7\n v_pk_mul_f32 %0, %0, %7"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6) : "v"(p7));
      } else if (MODE == 7) { // the step's mix: three packed instructions and five plain ones, a sequential add chain among them
        asm volatile("v_pk_mul_f32 %1, %1, %4\n v_add_f32 %0, %0, %6\n v_pk_mul_f32 %2, %2, %4\n v_add_f32 %0, %0, %7\n"
                     "v_pk_add_f32 %3, %3, %1\n v_add_f32 %0, %0, %6\n v_mul_f32 %0, %5, %0\n v_add_f32 %0, %0, %7"
                     : "+v"(a0), "+v"(p1), "+v"(p2), "+v"(p3) : "s"(sp), "s"(s0), "v"(a1), "v"(a2));
      } else if (MODE == 8) { // dependent mul+add recurrence (BU / AU): x = t + r*x
        asm volatile("v_mul_f32 %0, %2, %0\n v_add_f32 %0, %1, %0\n v_mul_f32 %0, %2, %0\n v_add_f32 %0, %1, %0\n"
                     "v_mul_f32 %0, %2, %0\n v_add_f32 %0, %1, %0\n v_mul_f32 %0, %2, %0\n v_add_f32 %0, %1, %0"
                     : "+v"(a0) : "v"(a1), "s"(s0));
      } else if (MODE == 5) {
        asm volatile("v_mul_f32 %0, %2, %0\n v_mul_f32 %1, %2, %1\n v_mul_f32 %0, %2, %0\n v_mul_f32 %1, %2, %1\n"
                     "v_mul_f32 %0, %2, %0\n v_mul_f32 %1, %2, %1\n v_mul_f32 %0, %2, %0\n v_mul_f32 %1, %2, %1"
                     : "+v"(a0), "+v"(a1) : "s"(s0));
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(p0 + p1 + p2 + p3 + p4 + p5 + p6 + p7) + r0 + r1 + r2 + r3;
}

template <int MODE> void run(const char* name, int wavesPerSimd)
{
  const int nCU = 256, iters = 4000;
  const int blocks = nCU * 4 * wavesPerSimd;
  float* out;
  hipMalloc(&out, (size_t)blocks * 64 * sizeof(float));
  hipMemset(out, 0, (size_t)blocks * 64 * sizeof(float));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, 10, 1.0f);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double instrPerWave = (double)iters * 32 * 8;
  printf("%-34s waves/SIMD=%d  %.3f ms  ns per instr per SIMD = %.3f\n", name, wavesPerSimd, ms,
         ms * 1e6 / (instrPerWave * wavesPerSimd));
  hipFree(out);
}

int main()
{
  for (int w : {1, 2, 4}) run<0>("v_mul_f32 independent", w);
  for (int w : {1, 2, 4}) run<1>("v_mul_f32 dependent chain", w);
  for (int w : {1, 2, 4}) run<5>("v_mul_f32 two chains interleaved", w);
  for (int w : {1, 2, 4}) run<2>("v_pk_mul_f32 independent", w);
  for (int w : {1, 2, 4}) run<3>("v_pk_mul_f32 dependent chain", w);
  for (int w : {1, 2, 4}) run<4>("v_readlane_b32 independent", w);
  for (int w : {1, 2, 4}) run<6>("v_pk_mul_f32 VGPR x VGPR indep", w);
  for (int w : {1, 2, 4}) run<7>("step-like mix (3 pk + 5 plain)", w);
  for (int w : {1, 2, 4}) run<8>("mul+add recurrence, dependent", w);
  return 0;
}
