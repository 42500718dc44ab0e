// Does hipExtAnyOrderLaunch let kernels of ONE stream overlap on gfx950?  (hip_ext.h says the flag is "not supported on AMD GFX9xx
// boards".)  Eight one-block kernels that each spin for 200 us go into one stream, once in order and once with the flag; in order
// they take 8 x 200 us, overlapped ~200 us.     hipcc --offload-arch=gfx950 -O2 anyorder_launch.hip -o anyorder_launch
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void spin(unsigned long long ticks, unsigned *out) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
  if (threadIdx.x == 0) atomicAdd(out, 1u);
}

int main() {
  unsigned *out;
  if (hipMalloc(&out, 4) != hipSuccess) return 1;
  hipMemset(out, 0, 4);
  hipStream_t st;
  hipStreamCreate(&st);
  unsigned long long ticks = 20000;  // 200 us of the 100 MHz clock
  void *args[] = {&ticks, &out};
  for (int flags = 0; flags <= 1; ++flags) {
    for (int rep = 0; rep < 3; ++rep) {
      hipStreamSynchronize(st);
      auto a = std::chrono::steady_clock::now();
      for (int i = 0; i < 8; ++i) {
        hipError_t e = hipExtLaunchKernel((const void *)spin, dim3(1), dim3(64), args, 0, st, nullptr, nullptr, flags);
        if (e != hipSuccess) {
          printf("launch failed: %s\n", hipGetErrorString(e));
          return 2;
        }
      }
      hipStreamSynchronize(st);
      auto b = std::chrono::steady_clock::now();
      printf("flags %d rep %d: 8 x 200 us kernels in one stream took %.0f us\n", flags, rep,
             std::chrono::duration<double, std::micro>(b - a).count());
    }
  }
  unsigned n = 0;
  hipMemcpy(&n, out, 4, hipMemcpyDeviceToHost);
  printf("kernels that ran: %u\n", n);
  return 0;
}
