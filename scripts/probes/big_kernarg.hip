#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { int n; float v[2047 + 1024]; };
__global__ void k(const Big b, float *out) { out[blockIdx.x] = b.v[blockIdx.x % (2047 + 1024)] + b.n; }
int main() {
  float *out; if (hipMalloc(&out, 4096 * 4) != hipSuccess) return 1;
  Big b; b.n = 1; for (int i = 0; i < 2047 + 1024; ++i) b.v[i] = i;
  hipLaunchKernelGGL(k, dim3(4096), dim3(64), 0, 0, b, out);
  hipError_t e = hipDeviceSynchronize();
  float h[4096]; (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  printf("%s: out[3000] = %f (want 3001)\n", hipGetErrorString(e), h[3000]);
  return 0;
}
