// Micro-benchmark: does the VGPR bank of the sources change the issue cost of VALU instructions on gfx950?
// Explicit registers; one wave per SIMD and two.  Build: hipcc -O3 --offload-arch=gfx950 valu_banks.hip -o valu_banks
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(x) x x x x
#define REP8(x) REP4(REP4(REP4(x)))
template <int MODE> __global__ void k(float* out, int iters)
{
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) { // pk, both sources on banks {0,1}
      asm volatile(REP8("v_pk_mul_f32 v[20:21], v[24:25], v[28:29]\n v_pk_mul_f32 v[32:33], v[36:37], v[40:41]\n"
                        "v_pk_mul_f32 v[44:45], v[48:49], v[52:53]\n v_pk_mul_f32 v[56:57], v[60:61], v[64:65]\n")
                   ::: "v20", "v21", "v32", "v33", "v44", "v45", "v56", "v57");
    } else if (MODE == 1) { // pk, sources on different bank pairs
      asm volatile(REP8("v_pk_mul_f32 v[20:21], v[24:25], v[30:31]\n v_pk_mul_f32 v[32:33], v[36:37], v[42:43]\n"
                        "v_pk_mul_f32 v[44:45], v[48:49], v[54:55]\n v_pk_mul_f32 v[56:57], v[60:61], v[66:67]\n")
                   ::: "v20", "v21", "v32", "v33", "v44", "v45", "v56", "v57");
    } else if (MODE == 2) { // plain, sources on the same bank
      asm volatile(REP8("v_add_f32 v20, v24, v28\n v_add_f32 v32, v36, v40\n v_add_f32 v44, v48, v52\n v_add_f32 v56, v60, v64\n")
                   ::: "v20", "v32", "v44", "v56");
    } else if (MODE == 3) { // plain, sources on different banks
      asm volatile(REP8("v_add_f32 v20, v24, v29\n v_add_f32 v32, v36, v41\n v_add_f32 v44, v48, v53\n v_add_f32 v56, v60, v65\n")
                   ::: "v20", "v32", "v44", "v56");
    } else if (MODE == 4) { // pk with an SGPR pair
      asm volatile(REP8("v_pk_mul_f32 v[20:21], s[20:21], v[28:29]\n v_pk_mul_f32 v[32:33], s[22:23], v[40:41]\n"
                        "v_pk_mul_f32 v[44:45], s[24:25], v[52:53]\n v_pk_mul_f32 v[56:57], s[26:27], v[64:65]\n")
                   ::: "v20", "v21", "v32", "v33", "v44", "v45", "v56", "v57");
    } else if (MODE == 5) { // the BU recurrence as compiled: v_mul (SGPR x VGPR) -> v_add, dependent, pk ops between
      asm volatile(REP8("v_mul_f32 v0, s28, v20\n v_add_f32 v21, v13, v0\n v_mul_f32 v0, s27, v21\n"
                        "v_pk_mul_f32 v[52:53], v[52:53], v[212:213]\n v_add_f32 v20, v12, v0\n"
                        "v_pk_mul_f32 v[10:11], s[90:91], v[52:53]\n")
                   ::: "v0", "v20", "v21", "v52", "v53", "v10", "v11");
    } else if (MODE == 6) { // dependent mul/add only
      asm volatile(REP8("v_mul_f32 v0, s28, v20\n v_add_f32 v21, v13, v0\n v_mul_f32 v0, s27, v21\n v_add_f32 v20, v12, v0\n")
                   ::: "v0", "v20", "v21");
    } else if (MODE == 7) { // dependent pk chain through one register pair
      asm volatile(REP8("v_pk_add_f32 v[20:21], v[20:21], v[30:31]\n v_pk_mul_f32 v[20:21], v[20:21], v[34:35]\n"
                        "v_pk_add_f32 v[20:21], v[20:21], v[38:39]\n v_pk_mul_f32 v[20:21], v[20:21], v[42:43]\n")
                   ::: "v20", "v21");
    } else if (MODE == 8) { // plain op consuming one half of a fresh pk result
      asm volatile(REP8("v_pk_mul_f32 v[20:21], v[24:25], v[30:31]\n v_add_f32 v24, v21, v40\n"
                        "v_pk_mul_f32 v[22:23], v[24:25], v[34:35]\n v_add_f32 v25, v22, v41\n")
                   ::: "v20", "v21", "v22", "v23", "v24", "v25");
    }
  }
  if (iters < 0) out[threadIdx.x] = 1.f;
}

template <int MODE> void run(const char* name, int wavesPerSimd, int perIter)
{
  const int nCU = 256, iters = 4000; perIter *= 8;
  const int blocks = nCU * 4 * wavesPerSimd;
  float* out;
  (void)hipMalloc(&out, 4096);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, 10);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double instrPerWave = (double)iters * perIter;
  printf("%-52s waves/SIMD=%d  %.3f ms  ns per instr per SIMD = %.3f  per wave = %.3f\n", name, wavesPerSimd, ms,
         ms * 1e6 / (instrPerWave * wavesPerSimd), ms * 1e6 / instrPerWave);
  (void)hipFree(out);
}

int main()
{
  for (int w : {1, 2}) run<0>("pk, VGPR sources on the same bank pair", w, 32);
  for (int w : {1, 2}) run<1>("pk, VGPR sources on different bank pairs", w, 32);
  for (int w : {1, 2}) run<2>("plain, sources on the same bank", w, 32);
  for (int w : {1, 2}) run<3>("plain, sources on different banks", w, 32);
  for (int w : {1, 2}) run<4>("pk, SGPR pair x VGPR pair", w, 32);
  for (int w : {1, 2}) run<5>("BU recurrence as compiled (4 plain dep + 2 pk)", w, 48);
  for (int w : {1, 2}) run<6>("dependent mul/add only", w, 32);
  for (int w : {1, 2}) run<7>("dependent pk chain", w, 32);
  for (int w : {1, 2}) run<8>("plain consumes half of a fresh pk result", w, 32);
  return 0;
}
