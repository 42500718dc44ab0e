// Do v_mfma_f32_16x16x4_f32 and ordinary VALU instructions of the waves of one SIMD overlap on gfx950, or do their issue cycles
// add up?  (The residual pass spends 16 MFMA = 512 matrix-pipe cycles and ~252 VALU = 1008 issue cycles per 64-pixel step;
// SQ_VALU_MFMA_COEXEC_CYCLES reads 0 for it.)  Three kernels, 4 waves per SIMD (256 blocks of 1024 threads), the same loop count:
// MFMA only, VALU only, both interleaved.  If the third takes the sum of the first two, nothing overlaps.
//   hipcc --offload-arch=gfx950 -O3 mfma_valu_coexec.hip -o mfma_valu_coexec
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE>  // 1: MFMA, 2: VALU, 3: both
__global__ __launch_bounds__(1024) void k(float *out, int iters, float seed) {
  v4f acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  float a = seed + threadIdx.x, b = seed * 0.5f;
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = seed * (i + 1);
  // (MODE 3: the odd waves of a SIMD start with the VALU half so that the four waves are not all in their matrix half at once)
  const bool swapped = MODE == 3 && ((threadIdx.x >> 8) & 1);  // (waves w and w + 4 of a 16-wave block share a SIMD)
  if (swapped) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], 1.0000001f, 0.5f);
    asm volatile("" ::: "memory");
  }
  for (int it = 0; it < iters; ++it) {
    if (MODE & 1) {
#pragma unroll
      for (int m = 0; m < 8; ++m) {  // 16 MFMAs, two independent chains
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, a, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, b, acc1, 0, 0, 0);
      }
    }
    if (MODE & 2) {
#pragma unroll
      for (int r = 0; r < 16; ++r)  // 256 independent-ish fmas (16 chains)
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], 1.0000001f, 0.5f);
    }
    asm volatile("" ::: "memory");
  }
  float s = acc0.x + acc0.y + acc0.z + acc0.w + acc1.x + acc1.y + acc1.z + acc1.w;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
float run(float *out, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, out, iters, 1.0f);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, out, iters, 1.0f);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float *out;
  if (hipMalloc(&out, 1024 * 256 * 4) != hipSuccess) return 1;
  const int iters = 2000;
  const float m = run<1>(out, iters), v = run<2>(out, iters), b = run<3>(out, iters);
  // per iteration and wave: 16 MFMA (x 32 cycles = 512) and 256 VALU (x 4 cycles = 1024); 4 waves per SIMD
  printf("MFMA only %.3f ms, VALU only %.3f ms, both %.3f ms (sum %.3f, max %.3f)\n", m, v, b, m + v, m > v ? m : v);
  printf("cycles per iteration and SIMD at 2.4 GHz: MFMA %.0f (4 waves x 512 = 2048 if the pipe is the bound), VALU %.0f (4096), both %.0f\n",
         m * 1e-3 * 2.4e9 / iters, v * 1e-3 * 2.4e9 / iters, b * 1e-3 * 2.4e9 / iters);
  return 0;
}
