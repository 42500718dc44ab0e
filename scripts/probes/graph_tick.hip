// Is a hipGraph a cheaper way to issue one tick of a single pair than two plain launches?  A tick of dvo_amd_match() is k_tick_small
// (896 bytes of by-value arguments that change every tick: the pose, the precision, the tick number) followed by k_finalize_small
// (520 bytes) on the same stream, and the host spinning on pinned memory for the record the second kernel publishes.  This probe
// runs that shape -- two small dependent kernels with argument blocks of those sizes, a tagged word in pinned host memory, the host
// polling it -- (a) as two hipLaunchKernelGGL calls per tick and (b) as ONE instantiated two-node graph whose kernel-node
// parameters are replaced every tick (hipGraphExecKernelNodeSetParams x 2) and launched with hipGraphLaunch, and prints the
// time per tick of each.
// build: hipcc --offload-arch=gfx950 -O2 scripts/probes/graph_tick.hip -o scripts/probes/bin/graph_tick
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

struct ArgsA {
  unsigned seq;
  unsigned pad[223];  // 896 bytes
};
struct ArgsB {
  unsigned seq;
  unsigned *out;  // pinned host word
  unsigned pad[126];  // 520 bytes
};
static_assert(sizeof(ArgsA) == 896, "");
static_assert(sizeof(ArgsB) == 520, "");

__global__ __launch_bounds__(256) void k_a(const ArgsA a, float *scratch) {
  // a few hundred cycles of work per block, 19 blocks: the size of a level-3 pass
  float v = (float)a.seq + (float)threadIdx.x;
  for (int i = 0; i < 64; ++i) v = v * 1.0001f + (float)a.pad[i & 127];
  scratch[blockIdx.x * 256 + threadIdx.x] = v;
}
__global__ __launch_bounds__(512) void k_b(const ArgsB b, const float *scratch) {
  __shared__ float s[512];
  s[threadIdx.x] = scratch[threadIdx.x];
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.0f;
    for (int i = 0; i < 512; i += 64) t += s[i];
    if (t == 12345.678f) b.out[1] = 1u;  // (keeps the loads)
    __hip_atomic_store(b.out, b.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

#define CHECK(x)                                                                       \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      std::printf("%s failed: %s\n", #x, hipGetErrorString(e_));                       \
      return 1;                                                                        \
    }                                                                                  \
  } while (0)

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv) {
  const int n = argc > 1 ? std::atoi(argv[1]) : 20000;
  hipStream_t st;
  CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  float *scratch;
  CHECK(hipMalloc(&scratch, sizeof(float) * 256 * 19));
  unsigned *out_host, *out_dev;
  CHECK(hipHostMalloc((void **)&out_host, 64, hipHostMallocMapped | hipHostMallocCoherent));
  CHECK(hipHostGetDevicePointer((void **)&out_dev, out_host, 0));
  out_host[0] = 0;
  ArgsA a;
  ArgsB b;
  std::memset(&a, 0, sizeof(a));
  std::memset(&b, 0, sizeof(b));
  b.out = out_dev;
  unsigned seq = 0;
  auto wait_for = [&](unsigned s) -> bool {
    const double t0 = now_us();
    while (__atomic_load_n(out_host, __ATOMIC_ACQUIRE) != s) {
      __builtin_ia32_pause();
      if (now_us() - t0 > 2e6) return false;
    }
    return true;
  };
  // (a) two launches per tick
  for (int pass = 0; pass < 2; ++pass) {
    const double t0 = now_us();
    for (int i = 0; i < n; ++i) {
      a.seq = b.seq = ++seq;
      hipLaunchKernelGGL(k_a, dim3(19), dim3(256), 0, st, a, scratch);
      hipLaunchKernelGGL(k_b, dim3(1), dim3(512), 0, st, b, (const float *)scratch);
      if (!wait_for(seq)) {
        std::printf("timeout (plain launches)\n");
        return 1;
      }
    }
    if (pass) std::printf("two plain launches per tick:                       %.2f us per tick\n", (now_us() - t0) / n);
  }
  CHECK(hipStreamSynchronize(st));
  // (b) one two-node graph, parameters replaced every tick
  hipGraph_t graph;
  CHECK(hipGraphCreate(&graph, 0));
  void *pa[2] = {&a, &scratch};
  const float *scratch_c = scratch;
  void *pb[2] = {&b, &scratch_c};
  hipKernelNodeParams na, nb;
  std::memset(&na, 0, sizeof(na));
  std::memset(&nb, 0, sizeof(nb));
  na.func = (void *)k_a, na.gridDim = dim3(19), na.blockDim = dim3(256), na.kernelParams = pa;
  nb.func = (void *)k_b, nb.gridDim = dim3(1), nb.blockDim = dim3(512), nb.kernelParams = pb;
  hipGraphNode_t node_a, node_b;
  CHECK(hipGraphAddKernelNode(&node_a, graph, nullptr, 0, &na));
  CHECK(hipGraphAddKernelNode(&node_b, graph, &node_a, 1, &nb));
  hipGraphExec_t exec;
  CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  for (int pass = 0; pass < 2; ++pass) {
    const double t0 = now_us();
    for (int i = 0; i < n; ++i) {
      a.seq = b.seq = ++seq;
      CHECK(hipGraphExecKernelNodeSetParams(exec, node_a, &na));
      CHECK(hipGraphExecKernelNodeSetParams(exec, node_b, &nb));
      CHECK(hipGraphLaunch(exec, st));
      if (!wait_for(seq)) {
        std::printf("timeout (graph)\n");
        return 1;
      }
    }
    if (pass) std::printf("one two-node graph, parameters replaced every tick: %.2f us per tick\n", (now_us() - t0) / n);
  }
  // (c) the graph without replacing parameters (what a graph costs when nothing changes: not usable for a tick, a floor)
  {
    const double t0 = now_us();
    int done = 0;
    for (int i = 0; i < n; ++i) {
      CHECK(hipGraphLaunch(exec, st));
      ++done;
      if ((i & 63) == 63) CHECK(hipStreamSynchronize(st));
    }
    CHECK(hipStreamSynchronize(st));
    std::printf("the same graph relaunched unchanged, no host wait:   %.2f us per launch (%d launches)\n", (now_us() - t0) / n, done);
  }
  CHECK(hipGraphExecDestroy(exec));
  CHECK(hipGraphDestroy(graph));
  return 0;
}
