// How many workgroups of k_tick's shape does the GPU hold at once, and how long does a block slot stay empty between two of
// them?  Blocks of 256 threads with LDS_BYTES of LDS and ~NV live vector registers stay for T microseconds each (spinning on
// arithmetic, or sleeping); 16 x 1 024 of them.  Per block: first / last instruction time stamps -> the most blocks resident at
// once, the average over the middle half of the run, the run's length.        hipcc --offload-arch=gfx950 -O3 slot_turnover.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

// ticks: a block's stay; ticks_spread > 0: block b stays ticks + (hash(b) % ticks_spread) instead (uneven block lives)
template <int LDS_BYTES, int NV, bool SLEEP>
__global__ __launch_bounds__(256, 4) void spin(unsigned long long ticks, unsigned long long *stamps, float *sink, unsigned ticks_spread = 0) {
  __shared__ float lds[LDS_BYTES / 4];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (ticks_spread) ticks += (blockIdx.x * 2654435761u >> 8) % ticks_spread;
  float v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = (float)(threadIdx.x + i);
  lds[threadIdx.x] = v[3];
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
    if (SLEEP) {
      __builtin_amdgcn_s_sleep(32);
#pragma unroll
      for (int i = 0; i < NV; ++i) asm volatile("" : "+v"(v[i]));
    } else {
#pragma unroll
      for (int i = 0; i < NV; ++i) v[i] = __builtin_fmaf(v[i], 1.0000001f, 0.25f);
    }
  }
  float s = lds[(threadIdx.x * 7) & 255];
#pragma unroll
  for (int i = 0; i < NV; ++i) s += v[i];
  if (s == 12345.678f) sink[0] = s;
  __syncthreads();
  if (threadIdx.x == 0) stamps[2 * blockIdx.x] = t0, stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
}

template <int LDS_BYTES, int NV, bool SLEEP>
int run(double T, unsigned long long *stamps, float *sink, double spread = 0.0) {
  const int n = 16 * 1024;
  const unsigned long long ticks = (unsigned long long)(T * 100.0);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((spin<LDS_BYTES, NV, SLEEP>), dim3(n), dim3(256), 0, 0, ticks, stamps, sink, (unsigned)(spread * 100.0));
    if (spread > 0.0 && rep == 1) printf("(block lives T .. T + %.0f us, hashed) ", spread);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
  }
  std::vector<unsigned long long> h(2 * n);
  if (hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * 2 * n, hipMemcpyDeviceToHost) != hipSuccess) return 3;
  unsigned long long lo = ~0ull, hi = 0;
  double busy = 0.0;
  for (int i = 0; i < n; ++i) lo = std::min(lo, h[2 * i]), hi = std::max(hi, h[2 * i + 1]), busy += (double)(h[2 * i + 1] - h[2 * i]);
  std::vector<std::pair<unsigned long long, int>> ev;
  for (int i = 0; i < n; ++i) ev.emplace_back(h[2 * i], 1), ev.emplace_back(h[2 * i + 1], -1);
  std::sort(ev.begin(), ev.end());
  int cur = 0, most = 0;
  double mid_area = 0.0, mid_time = 0.0;
  for (size_t k = 0; k + 1 < ev.size(); ++k) {
    cur += ev[k].second;
    most = std::max(most, cur);
    const double a = (double)(ev[k].first - lo), b = (double)(ev[k + 1].first - lo), sp = (double)(hi - lo);
    if (a > 0.25 * sp && b < 0.75 * sp) mid_area += cur * (b - a), mid_time += b - a;
  }
  const double life = busy / n / 100.0, span = (double)(hi - lo) / 100.0, mid = mid_area / std::max(mid_time, 1.0);
  printf("LDS %5d B, %3d registers kept, %s, T = %4.1f us: block life %5.2f us, most resident %4d, middle half %4.0f, %d blocks in %6.1f us "
         "=> per block and slot %.2f us empty (slots = most resident)\n",
         LDS_BYTES, NV, SLEEP ? "sleeping" : "spinning", T, life, most, mid, n, span, span * most / n - life);
  return 0;
}

int main() {
  unsigned long long *stamps;
  float *sink;
  if (hipMalloc(&stamps, sizeof(unsigned long long) * 2 * 16 * 1024) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
  run<36592, 100, false>(18.0, stamps, sink);
  run<36592, 100, true>(18.0, stamps, sink);
  run<36592, 40, false>(18.0, stamps, sink);
  run<36592, 40, true>(18.0, stamps, sink);
  run<16384, 40, true>(18.0, stamps, sink);
  run<16384, 100, true>(18.0, stamps, sink);
  run<1024, 40, true>(18.0, stamps, sink);
  run<36592, 100, true>(5.0, stamps, sink);
  run<36592, 100, true>(36.0, stamps, sink);
  run<36592, 100, true>(7.0, stamps, sink, 15.0);  // uneven lives: 7 .. 22 us, as in a batch launch
  run<36592, 100, true>(14.5, stamps, sink);       // ... against even ones of the same mean
  return 0;
}
