// What core clock do the shaders really run at while the bench loads the GPU?  One wave per XCD-ish spins for ~40 ms and
// compares the shader clock counter (s_memtime: core cycles) with the constant 100 MHz counter (s_memrealtime).
// build: hipcc --offload-arch=gfx950 -O2 scripts/probes/clock_probe.hip -o scripts/probes/clock_probe
// usage: clock_probe [samples] [interval_ms]   (run next to `python bench.py --steps 400 ...`)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

__global__ void k_clock(unsigned long long *out, unsigned spin_realtime_ticks) {
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = r0;
  while (r1 - r0 < spin_realtime_ticks) {
    __builtin_amdgcn_s_sleep(16);
    r1 = __builtin_amdgcn_s_memrealtime();
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = c1 - c0;
    out[2 * blockIdx.x + 1] = r1 - r0;
  }
}

int main(int argc, char **argv) {
  const int samples = argc > 1 ? atoi(argv[1]) : 10;
  const int interval_ms = argc > 2 ? atoi(argv[2]) : 500;
  const int blocks = 8;
  unsigned long long *d = nullptr, h[2 * blocks];
  if (hipMalloc((void **)&d, sizeof(h)) != hipSuccess) return 1;
  for (int s = 0; s < samples; ++s) {
    hipLaunchKernelGGL(k_clock, dim3(blocks), dim3(64), 0, 0, d, 2000000u);  // 20 ms
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    double lo = 1e9, hi = 0;
    for (int b = 0; b < blocks; ++b) {
      const double mhz = (double)h[2 * b] / (double)h[2 * b + 1] * 100.0;
      lo = mhz < lo ? mhz : lo, hi = mhz > hi ? mhz : hi;
    }
    std::printf("sample %d: shader clock %.0f .. %.0f MHz over %d blocks (20 ms each)\n", s, lo, hi, blocks);
    std::fflush(stdout);
    std::this_thread::sleep_for(std::chrono::milliseconds(interval_ms));
  }
  return 0;
}
