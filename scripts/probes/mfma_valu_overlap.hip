// Do fp32 MFMA (v_mfma_f32_16x16x4_f32) and plain fp32 VALU work of DIFFERENT waves on one SIMD overlap on gfx950?
// Kernel A: every wave issues only VALU; kernel B: only MFMA; kernel C: even waves VALU, odd waves MFMA (same counts per
// wave as in A / B).  4 waves per SIMD, all CUs filled.  Overlap <=> time(C) ~ max(A, B) / 1 rather than (A + B) / 2.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
  const int wave = threadIdx.x >> 6;
  float a0 = threadIdx.x * 1e-3f + 1.0f, a1 = a0 + 0.5f, a2 = a0 + 0.25f, a3 = a0 + 0.125f;
  float b0 = a0 + 2.0f, b1 = a1 + 2.0f, b2 = a2 + 2.0f, b3 = a3 + 2.0f;
  v4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
  const float m = 1.0000001f;
  const bool do_valu = MODE == 0 || (MODE == 2 && (wave & 1) == 0);
  const bool do_mfma = MODE == 1 || (MODE == 2 && (wave & 1) == 1);
  for (int i = 0; i < iters; ++i) {
    if (do_valu) {  // 16 independent-ish v_mul_f32 (8 chains x 2)
      asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n"
                   "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7\n"
                   "v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n"
                   "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(m));
    }
    if (do_mfma) {  // 2 MFMAs on two accumulators
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, c1, 0, 0, 0);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + b0 + b1 + b2 + b3 + c0[0] + c1[1];
}

template <int MODE>
float run(float *d, int iters) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(256 * 4), dim3(256), 0, 0, d, 16);
  (void)hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(256 * 4), dim3(256), 0, 0, d, iters);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms;
  (void)hipEventElapsedTime(&ms, a, b);
  return ms;
}

int main() {
  float *d;
  (void)hipMalloc(&d, sizeof(float) * 256 * 4 * 256);
  const int iters = 100000;
  const float ta = run<0>(d, iters), tb = run<1>(d, iters), tc = run<2>(d, iters);
  printf("A all waves VALU (16 v_mul per iteration): %.3f ms\n", ta);
  printf("B all waves MFMA (2 v_mfma_f32_16x16x4_f32 per iteration): %.3f ms  -> %.1f ns per MFMA per wave\n", tb, tb * 1e6 / (2.0 * iters));
  printf("C half the waves of every SIMD VALU, the other half MFMA: %.3f ms\n", tc);
  printf("no overlap would give (A + B) / 2 = %.3f ms, full overlap max(A, B) / 2 = %.3f ms\n", (ta + tb) / 2, (ta > tb ? ta : tb) / 2);
  return 0;
}
