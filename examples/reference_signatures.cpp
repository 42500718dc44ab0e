// The reference's REAL signatures -- Eigen::Affine3d, cv::Mat -- through the adaptor's production branch
// (DVO_AMD_HAVE_EIGEN / DVO_AMD_HAVE_OPENCV of include/dvo_amd/dense_tracking.hpp):
//   RgbdImagePyramidPtr RgbdCameraPyramid::create(const cv::Mat& base_intensity, const cv::Mat& base_depth)   rgbd_image.h:135
//   bool DenseTracker::match(RgbdImagePyramid&, RgbdImagePyramid&, Eigen::Affine3d& transformation)           dense_tracking.h:156
//   bool DenseTracker::match(PointSelection&,   RgbdImagePyramid&, Eigen::Affine3d& transformation)           dense_tracking.h:157
//   cv::Mat DenseTracker::computeIntensityErrorImage(RgbdImagePyramid&, RgbdImagePyramid&, const AffineTransformd&, size_t)  :162
//   SurfacePyramid::convertRawDepthImageSse(const cv::Mat&, cv::Mat&, float)                                  surface_pyramid.h
// written the way dvo_ros/src/camera_dense_tracking.cpp:216-276 and dvo_benchmark/src/benchmark_slam.cpp:46-93 call them.
// Neither Eigen nor OpenCV exists in this image: tests/test_cpp_adaptor.py compiles this file against the TEST-ONLY mocks in
// tests/mock_include (a compile proof of the branch, -std=c++11 -Wall -Wextra -Werror, and -- the mocks being functional -- a
// run on the GPU); with the real libraries on the include path it compiles unchanged.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include <Eigen/Geometry>
#include <opencv2/core/core.hpp>

#include "dvo_amd/dense_tracking.hpp"

#if !defined(DVO_AMD_HAVE_EIGEN) || !defined(DVO_AMD_HAVE_OPENCV)
#error "this example is the Eigen / OpenCV branch of the adaptor: put Eigen and OpenCV (or tests/mock_include) on the include path"
#endif

static_assert(std::is_same<dvo::core::AffineTransformd, Eigen::Affine3d>::value, "AffineTransformd is Eigen::Affine3d (datatypes.h)");
static_assert(std::is_same<dvo::core::Matrix6d, Eigen::Matrix<double, 6, 6> >::value, "Matrix6d is Eigen's");
static_assert(std::is_same<decltype(std::declval<dvo::DenseTracker &>().match(std::declval<dvo::core::RgbdImagePyramid &>(),
                                                                               std::declval<dvo::core::RgbdImagePyramid &>(),
                                                                               std::declval<Eigen::Affine3d &>())),
                           bool>::value,
              "bool match(RgbdImagePyramid&, RgbdImagePyramid&, Eigen::Affine3d&)");
static_assert(std::is_same<decltype(std::declval<dvo::DenseTracker &>().computeIntensityErrorImage(
                               std::declval<dvo::core::RgbdImagePyramid &>(), std::declval<dvo::core::RgbdImagePyramid &>(),
                               std::declval<const dvo::core::AffineTransformd &>(), (size_t)0)),
                           cv::Mat>::value,
              "cv::Mat computeIntensityErrorImage(...)");

static cv::Mat read_mat(const char *path, int rows, int cols, int type) {
  cv::Mat m(rows, cols, type);
  FILE *f = std::fopen(path, "rb");
  const size_t bytes = (size_t)rows * (size_t)m.step;
  if (!f || std::fread(m.ptr<unsigned char>(0), 1, bytes, f) != bytes) {
    std::fprintf(stderr, "cannot read %s\n", path);
    std::exit(2);
  }
  std::fclose(f);
  return m;
}

int main(int argc, char **argv) {
  if (argc < 11) {
    std::fprintf(stderr, "usage: %s w h fx fy ox oy ref_gray8 ref_depth16 cur_gray8 cur_depth16\n", argv[0]);
    return 2;
  }
  const int w = std::atoi(argv[1]), h = std::atoi(argv[2]);
  dvo::core::IntrinsicMatrix K =
      dvo::core::IntrinsicMatrix::create((float)std::atof(argv[3]), (float)std::atof(argv[4]), (float)std::atof(argv[5]), (float)std::atof(argv[6]));
  dvo::core::RgbdCameraPyramid camera(w, h, K);

  // benchmark_slam.cpp:56-80 / camera_dense_tracking.cpp:219-243: grey -> CV_32F, raw depth -> metres with NaN = invalid
  dvo::core::RgbdImagePyramidPtr pyr[2];
  for (int k = 0; k < 2; ++k) {
    cv::Mat grey8 = read_mat(argv[7 + 2 * k], h, w, CV_8UC1), depth16 = read_mat(argv[8 + 2 * k], h, w, CV_16UC1);
    cv::Mat intensity(h, w, CV_32FC1), depth;
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) intensity.at<float>(y, x) = (float)grey8.at<unsigned char>(y, x);  // convertTo(CV_32F)
    dvo::core::SurfacePyramid::convertRawDepthImageSse(depth16, depth, 1.0f / 5000.0f);
    pyr[k] = camera.create(intensity, depth);  // create(const cv::Mat&, const cv::Mat&)
  }

  dvo::DenseTracker::Config cfg = dvo::DenseTracker::getDefaultConfig();
  cfg.LastLevel = 0;
  dvo::DenseTracker tracker(cfg);

  Eigen::Affine3d transformation;
  transformation.setIdentity();
  const bool ok = tracker.match(*pyr[0], *pyr[1], transformation);  // bool match(..., Eigen::Affine3d&)
  std::printf("success %d\n", ok ? 1 : 0);
  for (int r = 0; r < 4; ++r)
    std::printf("%.17g %.17g %.17g %.17g\n", transformation(r, 0), transformation(r, 1), transformation(r, 2), transformation(r, 3));

  // the PointSelection overload with the same Eigen out-parameter (dense_tracking.h:157)
  dvo::core::ValidPointAndGradientThresholdPredicate predicate;
  dvo::core::PointSelection selection(*pyr[0], predicate);
  Eigen::Affine3d again = Eigen::Affine3d::Identity();
  const bool ok2 = tracker.match(selection, *pyr[1], again);
  double diff = 0.0;
  for (int i = 0; i < 16; ++i) diff += (again.matrix().data()[i] - transformation.matrix().data()[i]) * (again.matrix().data()[i] - transformation.matrix().data()[i]);
  std::printf("selection success %d same %d\n", ok2 ? 1 : 0, diff == 0.0 ? 1 : 0);

  // the composition the callers do with the result (camera_dense_tracking.cpp:262-270): Affine3d products and inverses
  Eigen::Affine3d accumulated = Eigen::Affine3d::Identity();
  accumulated = accumulated * transformation.inverse();
  std::printf("accumulated_tx %.17g\n", accumulated(0, 3));

  const cv::Mat err = tracker.computeIntensityErrorImage(*pyr[0], *pyr[1], transformation, 1);  // cv::Mat, CV_32FC1
  double s = 0.0;
  for (int y = 0; y < err.rows; ++y)
    for (int x = 0; x < err.cols; ++x) s += err.at<float>(y, x);
  std::printf("error_image %d x %d type_is_32f %d sum %.9g\n", err.cols, err.rows, err.type() == CV_32FC1 ? 1 : 0, s);
  return 0;
}
