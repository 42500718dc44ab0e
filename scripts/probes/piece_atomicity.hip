// Is a 16-byte store of one lane seen whole by a 16-byte load?  A writer lane stores {a, b, c, tag} with one
// global_store_dwordx4 (system scope) into fine-grained device memory or pinned host memory; a reader (the same lane right
// behind the store / another workgroup / the host CPU) polls the piece with one 16-byte load until the tag shows up and
// checks the payload.  Prints the number of torn pieces per variant.
#include <hip/hip_runtime.h>
#include <emmintrin.h>
#include <cstdio>
#include <cstring>
#include <thread>
#include <atomic>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
#define GLOBAL __attribute__((address_space(1)))

__device__ __forceinline__ void st16(void *p, v4u v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"((GLOBAL void *)p), "v"(v) : "memory"); }
__device__ __forceinline__ v4u ld16(const void *p) {
  v4u v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"((const GLOBAL void *)p) : "memory");
  return v;
}
__device__ __forceinline__ v4u make(unsigned i, unsigned tag) { return v4u{tag * 2654435761u + i, ~(tag * 40503u + i), tag ^ (i << 8), tag}; }
__device__ __forceinline__ bool good(v4u v, unsigned i) {
  const v4u w = make(i, v.w);
  return v.x == w.x && v.y == w.y && v.z == w.z;
}

// variant 0: every lane writes its piece and reads it back at once
__global__ void same_lane(v4u *buf, int rounds, unsigned *torn, int wait_between) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned bad = 0;
  for (int r = 1; r <= rounds; ++r) {
    st16(buf + i, make(i, (unsigned)r));
    if (wait_between) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    v4u v = ld16(buf + i);
    while (v.w != (unsigned)r) v = ld16(buf + i);
    if (!good(v, i)) ++bad;
  }
  if (bad) atomicAdd(torn, bad);
}
// variant 1: block 0 writes, block 1 reads (different CUs, maybe different XCDs)
__global__ void two_blocks(v4u *buf, int rounds, unsigned *torn, volatile unsigned *ack) {
  const unsigned i = threadIdx.x;
  unsigned bad = 0;
  for (int r = 1; r <= rounds; ++r) {
    if (blockIdx.x == 0) {
      st16(buf + i, make(i, (unsigned)r));
      if (i == 0) while (__hip_atomic_load(ack, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)r) __builtin_amdgcn_s_sleep(2);
      __syncthreads();
    } else {
      v4u v = ld16(buf + i);
      while (v.w != (unsigned)r) v = ld16(buf + i);
      if (!good(v, i)) ++bad;
      __syncthreads();
      if (i == 0) __hip_atomic_store(const_cast<unsigned *>(ack), (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (bad) atomicAdd(torn, bad);
}
// variant 2: the GPU writes into pinned host memory, the CPU reads
__global__ void to_host(v4u *host_buf, int rounds, volatile unsigned *host_ack) {
  const unsigned i = threadIdx.x;
  for (int r = 1; r <= rounds; ++r) {
    st16(host_buf + i, make(i, (unsigned)r));
    if (i == 0) while (__hip_atomic_load(host_ack, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != (unsigned)r) __builtin_amdgcn_s_sleep(2);
    __syncthreads();
  }
}

static bool good_host(const unsigned u[4], unsigned i) {
  const unsigned tag = u[3];
  return u[0] == tag * 2654435761u + i && u[1] == ~(tag * 40503u + i) && u[2] == (tag ^ (i << 8));
}

int main() {
  unsigned *torn;
  hipMalloc(&torn, 4);
  for (int fine = 0; fine < 2; ++fine) {
    v4u *buf;
    if (fine) {
      if (hipExtMallocWithFlags((void **)&buf, 1 << 20, hipDeviceMallocFinegrained) != hipSuccess) { printf("no fine-grained memory\n"); continue; }
    } else {
      hipMalloc(&buf, 1 << 20);
    }
    for (int wait = 0; wait < 2; ++wait) {
      hipMemset(buf, 0, 1 << 20);
      hipMemset(torn, 0, 4);
      hipLaunchKernelGGL(same_lane, dim3(64), dim3(256), 0, 0, buf, 20000, torn, wait);
      unsigned h = 0;
      hipMemcpy(&h, torn, 4, hipMemcpyDeviceToHost);
      printf("%s memory, same lane, %s: %u torn of %d\n", fine ? "fine-grained" : "coarse-grained", wait ? "store waited for" : "load right behind the store", h, 64 * 256 * 20000);
    }
    unsigned *ack;
    hipMalloc(&ack, 4);
    hipMemset(ack, 0, 4);
    hipMemset(buf, 0, 1 << 20);
    hipMemset(torn, 0, 4);
    hipLaunchKernelGGL(two_blocks, dim3(2), dim3(128), 0, 0, buf, 200000, torn, ack);
    unsigned h = 0;
    hipMemcpy(&h, torn, 4, hipMemcpyDeviceToHost);
    printf("%s memory, writer block -> reader block: %u torn of %d\n", fine ? "fine-grained" : "coarse-grained", h, 128 * 200000);
    hipFree(ack);
    hipFree(buf);
  }
  {
    v4u *hb;
    unsigned *hack;
    hipHostMalloc((void **)&hb, 4096, hipHostMallocMapped | hipHostMallocCoherent);
    hipHostMalloc((void **)&hack, 64, hipHostMallocMapped | hipHostMallocCoherent);
    memset(hb, 0, 4096);
    *hack = 0;
    v4u *db;
    unsigned *dack;
    hipHostGetDevicePointer((void **)&db, hb, 0);
    hipHostGetDevicePointer((void **)&dack, hack, 0);
    const int rounds = 200000, n = 66;
    hipLaunchKernelGGL(to_host, dim3(1), dim3(128), 0, 0, db, rounds, dack);
    long bad = 0;
    for (int r = 1; r <= rounds; ++r) {
      for (int i = 0; i < n; ++i) {
        alignas(16) unsigned u[4];
        do {
          __asm__ __volatile__("" ::: "memory");
          _mm_store_si128((__m128i *)u, _mm_load_si128((const __m128i *)(hb + i)));
        } while (u[3] != (unsigned)r);
        if (!good_host(u, (unsigned)i)) ++bad;
      }
      __atomic_store_n(hack, (unsigned)r, __ATOMIC_RELEASE);
    }
    hipDeviceSynchronize();
    printf("pinned host memory, GPU writes -> CPU reads (16-byte loads): %ld torn of %ld\n", bad, (long)rounds * n);
  }
  return 0;
}
