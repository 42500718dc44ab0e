// Issue rate of packed fp32 VALU ops against plain ones on gfx950: every wave runs N independent-chain instructions of one
// kind; all CUs are filled with 4 waves per SIMD so that the time is set by the issue rate, not by dependent latency.
// Also checks that v_pk_mul_f32 / v_pk_add_f32 honour MODE.FP_ROUND (toward zero) like their scalar forms.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void rate(float *out, int iters) {
  float a0 = threadIdx.x * 1e-3f + 1.0f, a1 = a0 + 0.5f, a2 = a0 + 0.25f, a3 = a0 + 0.125f;
  float b0 = a0 + 2.0f, b1 = a1 + 2.0f, b2 = a2 + 2.0f, b3 = a3 + 2.0f;
  const float m = 1.0000001f;
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {  // 8 independent v_mul_f32
      asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n"
                   "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(m));
    } else if (KIND == 1) {  // 8 independent v_pk_mul_f32 (16 multiplies)
      v2 p0 = {a0, b0}, p1 = {a1, b1}, p2 = {a2, b2}, p3 = {a3, b3};
      v2 mm = {m, m};
      asm volatile("v_pk_mul_f32 %0, %4, %0\n v_pk_mul_f32 %1, %4, %1\n v_pk_mul_f32 %2, %4, %2\n v_pk_mul_f32 %3, %4, %3\n"
                   "v_pk_mul_f32 %0, %4, %0\n v_pk_mul_f32 %1, %4, %1\n v_pk_mul_f32 %2, %4, %2\n v_pk_mul_f32 %3, %4, %3\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(mm));
      a0 = p0.x, b0 = p0.y, a1 = p1.x, b1 = p1.y, a2 = p2.x, b2 = p2.y, a3 = p3.x, b3 = p3.y;
    } else {  // 8 independent v_fma_f32
      asm volatile("v_fma_f32 %0, %8, %0, %8\n v_fma_f32 %1, %8, %1, %8\n v_fma_f32 %2, %8, %2, %8\n v_fma_f32 %3, %8, %3, %8\n"
                   "v_fma_f32 %4, %8, %4, %8\n v_fma_f32 %5, %8, %5, %8\n v_fma_f32 %6, %8, %6, %8\n v_fma_f32 %7, %8, %7, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(m));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + b0 + b1 + b2 + b3;
}

__global__ void round_probe(float *out) {
  // 1 + 2^-24 * 1.5 style products: x*y with an inexact result; RTZ and RN differ in the last bit
  v2 x = {1.0000001f, 1.9999999f}, y = {1.0000001f, 1.9999999f};
  asm volatile("" : "+v"(x), "+v"(y));
  v2 rn = x * y;
  float rn_s = x.x * y.x;
  asm volatile("" : "+v"(rn), "+v"(rn_s));
  __builtin_amdgcn_s_setreg(1 | (1 << 11), 3);
  asm volatile("" : "+v"(x), "+v"(y));
  v2 tz;
  asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(tz) : "v"(x), "v"(y));
  float tz_s;
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(tz_s) : "v"(x.x), "v"(y.x));
  asm volatile("" : "+v"(tz), "+v"(tz_s));
  __builtin_amdgcn_s_setreg(1 | (1 << 11), 0);
  out[0] = rn.x, out[1] = rn.y, out[2] = rn_s, out[3] = tz.x, out[4] = tz.y, out[5] = tz_s;
}

template <int KIND>
float run(float *d, int iters) {
  hipEvent_t a, b;
  hipEventCreate(&a), hipEventCreate(&b);
  hipLaunchKernelGGL(rate<KIND>, dim3(256 * 4), dim3(256), 0, 0, d, 16);
  hipEventRecord(a);
  hipLaunchKernelGGL(rate<KIND>, dim3(256 * 4), dim3(256), 0, 0, d, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms;
}

int main() {
  float *d;
  hipMalloc(&d, sizeof(float) * 256 * 4 * 256);
  const int iters = 200000;
  const float t0 = run<0>(d, iters), t1 = run<1>(d, iters), t2 = run<2>(d, iters);
  const double instr = 8.0 * iters;  // per wave
  printf("v_mul_f32   : %.3f ms  (%.2f ns per wave-instruction)\n", t0, t0 * 1e6 / instr);
  printf("v_pk_mul_f32: %.3f ms  (%.2f ns per wave-instruction, 2 multiplies each)\n", t1, t1 * 1e6 / instr);
  printf("v_fma_f32   : %.3f ms  (%.2f ns per wave-instruction)\n", t2, t2 * 1e6 / instr);
  hipLaunchKernelGGL(round_probe, dim3(1), dim3(64), 0, 0, d);
  float h[6];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("round: RN pk {%.9g, %.9g} scalar %.9g | RTZ pk {%.9g, %.9g} scalar %.9g\n", h[0], h[1], h[2], h[3], h[4], h[5]);
  return 0;
}
