// What does one tick of a device-side Gauss-Newton loop cost before any pixel work?  A persistent grid of G blocks; per tick the
// leader (block 0) publishes a 104-byte work item under a generation number, the first n_active blocks each write a 416-byte
// record through to memory and count themselves in, the leader waits for them, sums the records, runs `serial_ops` dependent
// fp64 operations on one lane (the stand-in for exp / log / compose / 6x6 solve) and publishes the next item.
// Every wait is bounded (s_memrealtime); a timeout ends the kernel and is reported.
// build: hipcc --offload-arch=gfx950 -O2 scripts/probes/loop_barrier.hip -o scripts/probes/loop_barrier
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

struct Ctrl {
  unsigned gen;
  unsigned pad0[31];
  unsigned arrived;
  unsigned pad1[31];
  unsigned item[32];
  unsigned status;  // 1: a wait timed out
};

#define AGENT __HIP_MEMORY_SCOPE_AGENT
__device__ __forceinline__ unsigned ld_agent(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, AGENT); }
__device__ __forceinline__ void st_agent(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, AGENT); }
__device__ __forceinline__ void st_agent_f(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, AGENT); }

__global__ __launch_bounds__(256, 2) void k_loop(Ctrl *ctrl, float *recs, int n_ticks, int n_active, int serial_ops,
                                                 unsigned timeout, int sleep_idle, double *out) {
  __shared__ unsigned sh_item[32];
  __shared__ int sh_stop;
  __shared__ double sh_sum[256];
  const int t = threadIdx.x;
  const bool leader = blockIdx.x == 0;
  double state = 1.0;
  if (leader && t == 0) {
    for (int i = 0; i < 26; ++i) st_agent(&ctrl->item[i], (unsigned)i);
    __builtin_amdgcn_s_waitcnt(0);
    st_agent(&ctrl->gen, 1u);
  }
  for (int tick = 0; tick < n_ticks; ++tick) {
    if (t == 0) {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      int stop = 0;
      while (ld_agent(&ctrl->gen) != (unsigned)tick + 1u) {
        if ((int)blockIdx.x >= n_active) { if (sleep_idle >= 32) __builtin_amdgcn_s_sleep(32); else __builtin_amdgcn_s_sleep(8); } else __builtin_amdgcn_s_sleep(1);
        if (__builtin_amdgcn_s_memrealtime() - t0 > timeout) { stop = 1; break; }
      }
      sh_stop = stop;
    }
    __syncthreads();
    if (sh_stop) { if (t == 0) st_agent(&ctrl->status, 1u); return; }
    if (t < 26) sh_item[t] = ld_agent(&ctrl->item[t]);
    __syncthreads();
    const unsigned salt = sh_item[3];
    if ((int)blockIdx.x < n_active) {
      if (t < 104) st_agent_f(recs + (size_t)blockIdx.x * 104 + t, (float)(salt & 7u) + (float)t);
      __builtin_amdgcn_s_waitcnt(0);  // every store of this lane has been acknowledged
      __syncthreads();
      if (t == 0) __hip_atomic_fetch_add(&ctrl->arrived, 1u, __ATOMIC_RELAXED, AGENT);
    }
    if (leader) {
      if (t == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        int stop = 0;
        const unsigned want = (unsigned)(tick + 1) * (unsigned)n_active;
        while (ld_agent(&ctrl->arrived) != want) {
          __builtin_amdgcn_s_sleep(1);
          if (__builtin_amdgcn_s_memrealtime() - t0 > timeout) { stop = 1; break; }
        }
        sh_stop = stop;
      }
      __syncthreads();
      if (sh_stop) { if (t == 0) st_agent(&ctrl->status, 1u), st_agent(&ctrl->gen, 0xFFFFFFFFu); return; }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      double s = 0.0;
      for (int i = t; i < n_active * 104; i += 256) s += (double)recs[i];
      sh_sum[t] = s;
      __syncthreads();
      if (t == 0) {
        double tot = 0.0;
        for (int i = 0; i < 256; ++i) tot += sh_sum[i];
        double x = state + tot * 1e-9;
        for (int i = 0; i < serial_ops; ++i) x = x * 1.0000001 + 1e-7;  // dependent multiply + add (not contracted: -ffp-contract=off)
        state = x;
        for (int i = 0; i < 26; ++i) st_agent(&ctrl->item[i], (unsigned)(tick + i));
        __builtin_amdgcn_s_waitcnt(0);
        st_agent(&ctrl->gen, (unsigned)tick + 2u);
      }
    }
  }
  if (leader && t == 0) out[0] = state;
}

int main(int argc, char **argv) {
  Ctrl *ctrl;
  float *recs;
  double *out;
  if (hipMalloc((void **)&ctrl, sizeof(Ctrl)) != hipSuccess) return 1;
  if (hipMalloc((void **)&recs, 1024 * 104 * 4) != hipSuccess) return 1;
  if (hipMalloc((void **)&out, 8) != hipSuccess) return 1;
  const int n_ticks = 2000;
  const int grids[] = {256, 512};
  const int actives[] = {19, 150, 300};
  const int serials[] = {0, 1000, 3000};
  const int sleeps[] = {8, 32};
  for (int g : grids)
    for (int a : actives)
      for (int s : serials)
        for (int sl : sleeps) {
          if (a > g) continue;
          (void)hipMemset(ctrl, 0, sizeof(Ctrl));
          (void)hipDeviceSynchronize();
          const auto t0 = std::chrono::steady_clock::now();
          hipLaunchKernelGGL(k_loop, dim3(g), dim3(256), 0, 0, ctrl, recs, n_ticks, a, s, 2000000u /* 20 ms */, sl, out);
          if (hipDeviceSynchronize() != hipSuccess) { std::printf("launch failed\n"); return 2; }
          const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
          Ctrl h;
          (void)hipMemcpy(&h, ctrl, sizeof(h), hipMemcpyDeviceToHost);
          std::printf("grid %3d active %3d serial_ops %4d idle_sleep %2d: %.2f us per tick%s\n", g, a, s, sl, us / n_ticks,
                      h.status ? "  (TIMED OUT)" : "");
          std::fflush(stdout);
          if (h.status) return 3;
        }
  return 0;
}
