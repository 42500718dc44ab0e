// v_mfma_f32_4x4x1_16b_f32 on gfx950: (a) which lane / register holds D[block][i][j], which lane supplies A[block][i], B[block][j];
// (b) issue cost next to v_mfma_f32_16x16x4_f32 (the Gram matrix of a 16-vector needs 9 of its 16 4x4 chunk pairs: 36
// instructions of the 16-block form per 64 points instead of 16 of the 16x16x4 form).
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize mfma_4x4_blocks.hip -o mfma_4x4_blocks
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void layout(float *out) {
  const int l = threadIdx.x;
  // A = 1000 + lane, B = 1 for lane l only if l == probe ... simpler: A[l] = l + 1, B[l] = 100 * (l + 1): D = A * B uniquely
  // identifies the (A lane, B lane) pair: D = (la + 1) * 100 * (lb + 1)
  v4f d = {0, 0, 0, 0};
  d = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(l + 1), 100.0f * (float)(l + 1), d, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = d[r];
}

template <int MODE>  // 0: 16 x 16x16x4, 1: 36 x 4x4x1_16b
__global__ __launch_bounds__(256) void timing(float *out, int iters, float seed) {
  v4f acc[9];
  for (int i = 0; i < 9; ++i) acc[i] = v4f{0, 0, 0, 0};
  float x[4] = {seed + threadIdx.x, seed * 0.5f, seed * 0.25f, seed * 2.0f};
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[0], x[0], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[1], x[1], acc[1], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        int p = 0;
#pragma unroll
        for (int ci = 0; ci < 4; ++ci)
#pragma unroll
          for (int cj = ci; cj < 4; ++cj) {
            if (ci == 3) continue;
            acc[p] = __builtin_amdgcn_mfma_f32_4x4x1f32(x[ci], x[cj], acc[p], 0, 0, 0);
            ++p;
          }
      }
    }
    asm volatile("" ::: "memory");
  }
  float s = 0;
  for (int i = 0; i < 9; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
float run(float *out, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(timing<MODE>, dim3(1024), dim3(256), 0, 0, out, iters, 1.0f);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(timing<MODE>, dim3(1024), dim3(256), 0, 0, out, iters, 1.0f);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float *out;
  if (hipMalloc(&out, 1024 * 256 * 4) != hipSuccess) return 1;
  hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, out);
  float h[256];
  if (hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 2;
  // decode: D = (la + 1) * 100 * (lb + 1); find for a few (lane, reg) the (la, lb)
  int ok_model = 1;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const float d = h[l * 4 + r];
      int fa = -1, fb = -1;
      for (int la = 0; la < 64 && fa < 0; ++la)
        for (int lb = 0; lb < 64; ++lb)
          if (d == (float)(la + 1) * 100.0f * (float)(lb + 1)) {
            // products are not unique in general; prefer the candidate inside the same block of four lanes
            if (la / 4 == l / 4 && lb / 4 == l / 4) { fa = la, fb = lb; break; }
          }
      // model: D register r of lane l = A[lane (l & ~3) + r] * B[lane l]  (block l / 4, row r, column l % 4)
      const float want = (float)((l & ~3) + r + 1) * 100.0f * (float)(l + 1);
      if (d != want) ok_model = 0;
      if (l < 8) printf("lane %2d reg %d: D = %9.0f  (A lane %d, B lane %d)\n", l, r, d, fa, fb);
    }
  printf("model 'register r of lane l = A[block l/4][row r] * B[block l/4][col l%%4], A from lane 4b+i, B from lane 4b+j': %s\n",
         ok_model ? "HOLDS for all 64 lanes" : "does NOT hold");
  const int iters = 4000;
  const float a = run<0>(out, iters), b = run<1>(out, iters);
  printf("16 x v_mfma_f32_16x16x4_f32: %.3f ms (%.0f cycles per iteration and SIMD at 2.4 GHz; 4 waves x 16 x 32 = 2048)\n", a, a * 1e-3 * 2.4e9 / iters);
  printf("36 x v_mfma_f32_4x4x1_16b_f32: %.3f ms (%.0f cycles; 4 waves x 36 x 8 = 1152)\n", b, b * 1e-3 * 2.4e9 / iters);
  return 0;
}
