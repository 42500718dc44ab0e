// What does the per-tick host arithmetic of the Gauss-Newton driver cost on ONE lane of the GPU?  (Second unknown of a device-
// side loop, next to the hand-off floor of loop_barrier.hip.)  Thread 0 of one block runs, per tick, what dvo_tracker.cpp's
// process_loglik + process_residual + begin_iteration + make_kt run on the host, built from the library's own se3.h:
//   critical path (needed before the next residual pass can be described):
//     log(det P), 2x2 inverse, the 87 -> 27 rebuild of A and b, pivoted 6x6 LDL^T solve, exp, compose, K*T
//   deferrable (statistics): log(initial) for the prior, inverse + compose for initial(), two 4x4 matrices
// build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -I dvo_slam_amd/csrc scripts/probes/gn_lane_cost.hip -o scripts/probes/gn_lane_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include "se3.h"

using namespace dvo_amd;

struct State {
  double acc[87];
  double S[3];
  SE3 estimate, initial;
  double x[6];
  double sink[64];
  float kt[12];
};

__device__ void rebuild(const double *acc, const float P[4], double A[36], double b[6]) {
  const double p00 = P[0], p10 = P[1], p01 = P[2], p11 = P[3];
  const double pab = 0.5 * (p01 + p10);
  int t = 0;
  for (int r = 0; r < 6; ++r)
    for (int c = r; c < 6; ++c, ++t) {
      const double v = p00 * acc[t] + pab * acc[21 + t] + p11 * acc[42 + t];
      A[c * 6 + r] = v;
      A[r * 6 + c] = v;
    }
  for (int i = 0; i < 6; ++i) b[i] = -(p00 * acc[63 + i] + p10 * acc[75 + i] + p01 * acc[69 + i] + p11 * acc[81 + i]);
}

__global__ __launch_bounds__(256) void k_cost(State *st, int n_ticks, int with_deferred) {
  __shared__ State s;
  if (threadIdx.x == 0) {
    s = *st;
    for (int tick = 0; tick < n_ticks; ++tick) {
      // ---- critical path
      const float scale = 1.0f / (float)(70000 - 3);
      float cov[4] = {(float)(s.S[0] * (double)scale), (float)(s.S[1] * (double)scale), (float)(s.S[1] * (double)scale),
                      (float)(s.S[2] * (double)scale)};
      float P[4];
      inverse2x2f(cov, P);
      const float det = P[0] * P[3] - P[1] * P[2];
      const float ll = (float)(0.5 * 70000.0 * log((double)det) - 3.5 * s.sink[0]);
      double A[36], b[6], xn[6];
      rebuild(s.acc, P, A, b);
      solve_ldlt6(A, b, xn);
      const bool cont = inf_norm6(xn) > 5e-7;
      const SE3 inc = se3_exp(xn);
      s.estimate = se3_compose(inc, s.estimate);
      double T[16];
      se3_matrix(s.estimate, T);
      const float K[9] = {525.0f, 0.0f, 319.5f, 0.0f, 525.0f, 239.5f, 0.0f, 0.0f, 1.0f};
      for (int i = 0; i < 3; ++i)
        for (int c = 0; c < 4; ++c)
          s.kt[i * 4 + c] = (K[i * 3 + 0] * (float)T[c * 4 + 0] + K[i * 3 + 1] * (float)T[c * 4 + 1]) + K[i * 3 + 2] * (float)T[c * 4 + 2];
      s.sink[1] += (double)ll + (cont ? 1.0 : 0.0);
      // ---- deferrable
      if (with_deferred) {
        s.initial = se3_compose(se3_inverse(inc), s.initial);
        double xi[6];
        se3_log(s.initial, xi);
        double sq = 0.0;
        for (int i = 0; i < 6; ++i) sq += xi[i] * xi[i];
        double Ti[16];
        se3_matrix(s.initial, Ti);
        for (int i = 0; i < 16; ++i) s.sink[2 + i] = T[i] + Ti[i];
        for (int i = 0; i < 36; ++i) s.sink[20 + (i & 31)] += A[i];
        s.sink[60] = sq;
      }
      // keep the next tick's input moving (and the increment small): perturb the moments with this tick's increment
      for (int i = 0; i < 6; ++i) s.acc[63 + i] = s.acc[63 + i] * 0.5 + xn[i] * 1e-3;
    }
    *st = s;
  }
}

int main() {
  State h;
  // a well-conditioned system: A = sum over a few "pixels" of J^T J
  for (int i = 0; i < 87; ++i) h.acc[i] = 0.0;
  unsigned seed = 12345u;
  auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return (double)(seed >> 8) / 16777216.0 - 0.5; };
  for (int px = 0; px < 64; ++px) {
    double Ja[6], Jb[6];
    for (int i = 0; i < 6; ++i) Ja[i] = rnd() * 100.0, Jb[i] = rnd() * 100.0;
    const double r0 = rnd() * 0.01, r1 = rnd() * 0.01;
    int t = 0;
    for (int i = 0; i < 6; ++i) {
      for (int j = i; j < 6; ++j, ++t) h.acc[t] += Ja[i] * Ja[j], h.acc[21 + t] += Ja[i] * Jb[j] + Jb[i] * Ja[j], h.acc[42 + t] += Jb[i] * Jb[j];
      h.acc[63 + i] += Ja[i] * r0, h.acc[69 + i] += Ja[i] * r1, h.acc[75 + i] += Jb[i] * r0, h.acc[81 + i] += Jb[i] * r1;
    }
  }
  h.S[0] = 3.0, h.S[1] = 0.1, h.S[2] = 2.0;
  h.estimate = SE3::identity(), h.initial = SE3::identity();
  for (int i = 0; i < 64; ++i) h.sink[i] = 0.0;
  State *d;
  if (hipMalloc((void **)&d, sizeof(State)) != hipSuccess) return 1;
  const int n = 2000;
  for (int with_deferred = 0; with_deferred < 2; ++with_deferred)
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipMemcpy(d, &h, sizeof(State), hipMemcpyHostToDevice);
      (void)hipDeviceSynchronize();
      const auto t0 = std::chrono::steady_clock::now();
      hipLaunchKernelGGL(k_cost, dim3(1), dim3(256), 0, 0, d, n, with_deferred);
      if (hipDeviceSynchronize() != hipSuccess) return 2;
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      State out;
      (void)hipMemcpy(&out, d, sizeof(State), hipMemcpyDeviceToHost);
      std::printf("one lane, %s: %.2f us per tick   (check %.6g %.6g)\n", with_deferred ? "critical path + statistics" : "critical path only    ",
                  us / n, out.sink[1], out.estimate.t[0]);
    }
  // the same arithmetic on this host core, for scale
  {
    State s = h;
    const auto t0 = std::chrono::steady_clock::now();
    for (int tick = 0; tick < 200000; ++tick) {
      float cov[4] = {3.0f / 69997.0f, 0.1f / 69997.0f, 0.1f / 69997.0f, 2.0f / 69997.0f}, P[4];
      inverse2x2f(cov, P);
      double A[36], b[6], xn[6];
      const double p00 = P[0], p10 = P[1], p01 = P[2], p11 = P[3], pab = 0.5 * (p01 + p10);
      int t = 0;
      for (int r = 0; r < 6; ++r)
        for (int c = r; c < 6; ++c, ++t) A[c * 6 + r] = A[r * 6 + c] = p00 * s.acc[t] + pab * s.acc[21 + t] + p11 * s.acc[42 + t];
      for (int i = 0; i < 6; ++i) b[i] = -(p00 * s.acc[63 + i] + p10 * s.acc[75 + i] + p01 * s.acc[69 + i] + p11 * s.acc[81 + i]);
      solve_ldlt6(A, b, xn);
      const SE3 inc = se3_exp(xn);
      s.estimate = se3_compose(inc, s.estimate);
      s.initial = se3_compose(se3_inverse(inc), s.initial);
      double xi[6];
      se3_log(s.initial, xi);
      for (int i = 0; i < 6; ++i) s.acc[63 + i] = s.acc[63 + i] * 0.5 + xn[i] * 1e-3 + xi[i] * 1e-12;
    }
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    std::printf("host core, critical path + log: %.3f us per tick   (check %.6g)\n", us / 200000, s.estimate.t[0]);
  }
  return 0;
}
