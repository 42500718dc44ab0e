// Minimal caller written the way dvo_ros/src/camera_dense_tracking.cpp:243-276 uses the reference API.
// Reads two raw float32 frames (intensity, depth) per camera from files written by tests/test_cpp_adaptor.py and prints the
// estimated transformation.  Build: see tests/test_cpp_adaptor.py or INTEGRATION.md.
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <vector>

#include "dvo_amd/dense_tracking.hpp"

static std::vector<float> read_plane(const char *path, size_t n) {
  std::vector<float> v(n);
  FILE *f = std::fopen(path, "rb");
  if (!f || std::fread(v.data(), sizeof(float), n, f) != n) {
    std::fprintf(stderr, "cannot read %s\n", path);
    std::exit(2);
  }
  std::fclose(f);
  return v;
}

int main(int argc, char **argv) {
  if (argc < 11) {
    std::fprintf(stderr, "usage: %s w h fx fy ox oy ref_i ref_z cur_i cur_z\n", argv[0]);
    return 2;
  }
  const int w = std::atoi(argv[1]), h = std::atoi(argv[2]);
  const size_t n = (size_t)w * h;
  dvo::core::IntrinsicMatrix K =
      dvo::core::IntrinsicMatrix::create((float)std::atof(argv[3]), (float)std::atof(argv[4]), (float)std::atof(argv[5]), (float)std::atof(argv[6]));
  std::vector<float> ri = read_plane(argv[7], n), rz = read_plane(argv[8], n), ci = read_plane(argv[9], n), cz = read_plane(argv[10], n);

  dvo::core::RgbdCameraPyramid camera(w, h, K);
  dvo::core::RgbdImagePyramidPtr reference = camera.create(ri.data(), rz.data());
  dvo::core::RgbdImagePyramidPtr current = camera.create(ci.data(), cz.data());

  dvo::DenseTracker::Config cfg = dvo::DenseTracker::getDefaultConfig();
  cfg.LastLevel = 0;
  dvo::DenseTracker tracker(cfg);
  std::cerr << cfg << std::endl;

  dvo::DenseTracker::Result result;
  tracker.match(*reference, *current, result);
  const double *T = dvo::core::data(result.Transformation);
  std::printf("isnan %d loglik %.9g levels %zu\n", result.isNaN() ? 1 : 0, result.LogLikelihood, result.Statistics.Levels.size());
  for (int r = 0; r < 4; ++r) std::printf("%.17g %.17g %.17g %.17g\n", T[r], T[4 + r], T[8 + r], T[12 + r]);
  for (size_t l = 0; l < result.Statistics.Levels.size(); ++l)
    std::printf("level %zu iterations %zu termination %d valid %zu\n", result.Statistics.Levels[l].Id,
                result.Statistics.Levels[l].Iterations.size(), (int)result.Statistics.Levels[l].TerminationCriterion,
                result.Statistics.Levels[l].ValidPixels);
  return 0;
}
