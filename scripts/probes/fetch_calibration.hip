// What does FETCH_SIZE (rocprofv3 --pmc) count for the access widths k_tick uses?  MI355X_MICROARCH.md: on gfx950 it reports
// exactly half of the bytes of a wide (16 B per lane) coalesced streaming read and is uncalibrated for other widths.  k_tick reads
//   - the reference planes with 4-byte-per-lane streaming loads (256 B per wave instruction),
//   - the spilled residuals with 8-byte-per-lane streaming loads (512 B per wave instruction),
//   - the current planes with 16-byte and 8-byte GATHERS: lane l reads the pixel its point projects to and the pixel right of it
//     (two 16-byte loads 16 bytes apart, in two rows; the lanes of a wave land on ~64 neighbouring pixels).
// Every kernel below reads a buffer far beyond the 256 MiB Infinity Cache exactly once with one of these patterns; the known
// byte count is printed, FETCH_SIZE of the dispatch comes from the profiler:
//   rocprofv3 --pmc FETCH_SIZE -d OUT -o cal --output-format csv -- ./fetch_calibration
// factor = known bytes / (FETCH_SIZE x 1024).  Build: hipcc --offload-arch=gfx950 -O2 -o fetch_calibration fetch_calibration.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <class T>
__global__ void k_stream(const T *__restrict__ in, float *__restrict__ out, size_t n) {
  // grid-stride, one element per lane per trip: a wave instruction reads 64 consecutive elements
  float acc = 0.0f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const T v = in[i];
    acc += ((const float *)&v)[0];
  }
  if (acc == 12345.678f) out[0] = acc;  // (keeps the loads)
}

// k_tick's gather: pixel p of a w-wide image of 16-byte texels; lane reads texels (p, p + 1) of its row and of the next row.
// Points of a wave project to 64 neighbouring pixels (stride ~1 pixel), waves walk the image once.
__global__ void k_gather(const v4f *__restrict__ img, float *__restrict__ out, int w, int h) {
  float acc = 0.0f;
  const size_t n = (size_t)w * (h - 1) - 1;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const v4f a = img[i], b = img[i + 1], c = img[i + w], d = img[i + w + 1];
    acc += a.x + b.y + c.z + d.w;
  }
  if (acc == 12345.678f) out[0] = acc;
}

int main() {
  const size_t bytes = (size_t)3 << 30;  // 3 GiB: twelve times the Infinity Cache
  void *buf = nullptr;
  float *out = nullptr;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
  (void)hipMemset(buf, 0, bytes);
  (void)hipDeviceSynchronize();
  const dim3 grid(256 * 16), block(256);
  hipLaunchKernelGGL(k_stream<float>, grid, block, 0, 0, (const float *)buf, out, bytes / 4);
  (void)hipDeviceSynchronize();
  std::printf("k_stream<float>  (4 B per lane, streaming):  %zu bytes\n", bytes);
  hipLaunchKernelGGL(k_stream<v2f>, grid, block, 0, 0, (const v2f *)buf, out, bytes / 8);
  (void)hipDeviceSynchronize();
  std::printf("k_stream<float2> (8 B per lane, streaming):  %zu bytes\n", bytes);
  hipLaunchKernelGGL(k_stream<v4f>, grid, block, 0, 0, (const v4f *)buf, out, bytes / 16);
  (void)hipDeviceSynchronize();
  std::printf("k_stream<float4> (16 B per lane, streaming): %zu bytes\n", bytes);
  const int w = 8192, h = (int)(bytes / 16 / w);
  hipLaunchKernelGGL(k_gather, grid, block, 0, 0, (const v4f *)buf, out, w, h);
  const hipError_t e = hipDeviceSynchronize();
  std::printf("k_gather (four 16-B texels per lane, neighbouring pixels, every texel of the image touched 4x): %zu bytes of image\n",
              (size_t)w * h * 16);
  std::printf("%s\n", hipGetErrorString(e));
  return e == hipSuccess ? 0 : 1;
}
