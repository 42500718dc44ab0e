// Does v_mfma_f32_16x16x4_f32 honour MODE.FP_ROUND?  D = 1 + 0.75 ulp: RNE gives 1+2^-23, RTZ gives 1.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4 __attribute__((ext_vector_type(4)));
__global__ void probe(float *out, int rtz) {
  if (rtz) __builtin_amdgcn_s_setreg(1 | (1 << 11), 3);
  const int lane = threadIdx.x;
  float a = (lane >> 4) == 0 ? 1.0f : 0.0f;             // A[row][k=0] = 1
  float b = (lane >> 4) == 0 ? 0.75f * 1.1920929e-07f : 0.0f;  // B[k=0][col] = 0.75 * 2^-23
  asm volatile("" : "+v"(a), "+v"(b));
  v4 c = {1.0f, 1.0f, 1.0f, 1.0f};
  v4 d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  float v = 1.0f + b * a;  // VALU reference under the same mode (no contraction)
  asm volatile("" : "+v"(v));
  if (rtz) __builtin_amdgcn_s_setreg(1 | (1 << 11), 0);
  if (lane == 0) out[rtz * 2] = d[0], out[rtz * 2 + 1] = v;
}
int main() {
  float *d; hipMalloc(&d, 16);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 0);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 1);
  float h[4]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  printf("RN : mfma %.9g valu %.9g\nRTZ: mfma %.9g valu %.9g\n", h[0], h[1], h[2], h[3]);
  return 0;
}
