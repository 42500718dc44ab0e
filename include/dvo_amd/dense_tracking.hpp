// dense_tracking.hpp -- header-only C++ adaptor over the C ABI (include/dvo_amd.h).
//
// Re-declares, name for name, the part of dvo_core's interface that callers of the dense-tracking hot path use:
//   dvo::core::IntrinsicMatrix        dvo_core/include/dvo/core/intrinsic_matrix.h:33-64
//   dvo::core::RgbdCameraPyramid      dvo_core/include/dvo/core/rgbd_image.h:127-144
//   dvo::core::RgbdImagePyramid       dvo_core/include/dvo/core/rgbd_image.h:242-262
//   dvo::DenseTracker {Config, TerminationCriteria, IterationStats, LevelStats, Stats, Result, configure, match}
//                                     dvo_core/include/dvo/dense_tracking.h:39-213
// so that dvo_ros / dvo_slam / dvo_benchmark call sites (camera_dense_tracking.cpp:243-276, local_tracker.cpp:157-213,
// constraint_proposal_validator.cpp:132-166, benchmark_slam.cpp:352-547) compile against it unchanged apart from the
// include path.  Where <Eigen/Geometry> / <opencv2/core/core.hpp> are on the include path the Eigen and cv::Mat
// signatures of the reference are used; otherwise minimal stand-in value types with the same accessors are provided
// (this image has neither, which is what the CPU compile check in tests/ exercises).
//
// All compute happens in libdvo_amd.so on the GPU.  Errors that the reference signals with assert() are thrown as
// dvo::DvoAmdError; the failure modes the reference reports through its Result (NaN, TooFewConstraints) are preserved.
#ifndef DVO_AMD_DENSE_TRACKING_HPP_
#define DVO_AMD_DENSE_TRACKING_HPP_

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <memory>
#include <ostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../dvo_amd.h"

#if !defined(DVO_AMD_NO_EIGEN) && defined(__has_include)
#if __has_include(<Eigen/Geometry>)
#include <Eigen/Geometry>
#define DVO_AMD_HAVE_EIGEN 1
#endif
#endif
#if !defined(DVO_AMD_NO_OPENCV) && defined(__has_include)
#if __has_include(<opencv2/core/core.hpp>)
#include <opencv2/core/core.hpp>
#define DVO_AMD_HAVE_OPENCV 1
#endif
#endif

namespace dvo {

class DvoAmdError : public std::runtime_error {
 public:
  DvoAmdError(int status, const std::string &where)
      : std::runtime_error(where + ": " + dvo_amd_status_string(status) + " [" + dvo_amd_last_error() + "]"), status_(status) {}
  int status() const { return status_; }

 private:
  int status_;
};

namespace detail {
inline void check(int status, const char *where) {
  if (status != DVO_AMD_OK) throw DvoAmdError(status, where);
}
}  // namespace detail

namespace core {

#ifdef DVO_AMD_HAVE_EIGEN
typedef Eigen::Affine3d AffineTransformd;
typedef Eigen::Matrix<double, 6, 6> Matrix6d;
typedef Eigen::Matrix<double, 6, 1> Vector6d;
typedef Eigen::Matrix2d Matrix2d;
typedef Eigen::Vector2d Vector2d;
inline const double *data(const AffineTransformd &T) { return T.matrix().data(); }
inline double *data(AffineTransformd &T) { return T.matrix().data(); }
inline double *data(Matrix6d &m) { return m.data(); }
#else
// column-major stand-ins with the accessors the call sites use
template <int R, int C>
struct Mat {
  double m[R * C];
  Mat() { std::memset(m, 0, sizeof(m)); }
  double &operator()(int r, int c) { return m[c * R + r]; }
  double operator()(int r, int c) const { return m[c * R + r]; }
  double &operator()(int i) { return m[i]; }
  double operator()(int i) const { return m[i]; }
  double *data() { return m; }
  const double *data() const { return m; }
  void setZero() { std::memset(m, 0, sizeof(m)); }
  void setIdentity() {
    setZero();
    for (int i = 0; i < (R < C ? R : C); ++i) (*this)(i, i) = 1.0;
  }
  void setConstant(double v) { std::fill(m, m + R * C, v); }
  double sum() const {
    double s = 0;
    for (int i = 0; i < R * C; ++i) s += m[i];
    return s;
  }
};
typedef Mat<6, 6> Matrix6d;
typedef Mat<6, 1> Vector6d;
typedef Mat<2, 2> Matrix2d;
typedef Mat<2, 1> Vector2d;
struct AffineTransformd {
  Mat<4, 4> mat;
  AffineTransformd() { mat.setIdentity(); }
  Mat<4, 4> &matrix() { return mat; }
  const Mat<4, 4> &matrix() const { return mat; }
  void setIdentity() { mat.setIdentity(); }
  double &operator()(int r, int c) { return mat(r, c); }
  double operator()(int r, int c) const { return mat(r, c); }
};
inline const double *data(const AffineTransformd &T) { return T.mat.data(); }
inline double *data(AffineTransformd &T) { return T.mat.data(); }
inline double *data(Matrix6d &m) { return m.data(); }
#endif

// intrinsic_matrix.h:33-64
class IntrinsicMatrix {
 public:
  IntrinsicMatrix() : fx_(0), fy_(0), ox_(0), oy_(0) {}
  static IntrinsicMatrix create(float fx, float fy, float ox, float oy) {
    IntrinsicMatrix k;
    k.fx_ = fx, k.fy_ = fy, k.ox_ = ox, k.oy_ = oy;
    return k;
  }
  float fx() const { return fx_; }
  float fy() const { return fy_; }
  float ox() const { return ox_; }
  float oy() const { return oy_; }
  void invertOffset() { ox_ *= -1, oy_ *= -1; }
  void scale(float factor) { fx_ *= factor, fy_ *= factor, ox_ *= factor, oy_ *= factor; }

 private:
  float fx_, fy_, ox_, oy_;
};

class RgbdCameraPyramid;

// rgbd_image.h:242-262.  The reference extends a pyramid lazily, level by level; here the base planes are kept on the host
// until the first build(n), then the device pyramid is (re)built with max(n, levels so far) levels in one go.
class RgbdImagePyramid {
 public:
  typedef std::shared_ptr<RgbdImagePyramid> Ptr;

  RgbdImagePyramid(int width, int height, const IntrinsicMatrix &K, const float *intensity, const float *depth, int stride,
                   int device, double timestamp)
      : width_(width), height_(height), K_(K), device_(device), timestamp_(timestamp), handle_(nullptr), levels_(0) {
    intensity_.resize((size_t)width * height);
    depth_.resize((size_t)width * height);
    for (int y = 0; y < height; ++y) {
      std::memcpy(&intensity_[(size_t)y * width], intensity + (size_t)y * stride, sizeof(float) * width);
      std::memcpy(&depth_[(size_t)y * width], depth + (size_t)y * stride, sizeof(float) * width);
    }
  }
  ~RgbdImagePyramid() { dvo_amd_pyramid_release(handle_); }
  RgbdImagePyramid(const RgbdImagePyramid &) = delete;
  RgbdImagePyramid &operator=(const RgbdImagePyramid &) = delete;

  void compute(const size_t num_levels) { build(num_levels); }  // deprecated spelling kept by the reference (:252)

  void build(const size_t num_levels) {
    if ((size_t)levels_ >= num_levels && handle_) return;
    const int want = (int)std::max<size_t>(num_levels, (size_t)levels_);
    dvo_amd_pyramid *fresh = nullptr;
    detail::check(dvo_amd_pyramid_create(device_, intensity_.data(), depth_.data(), width_, height_, width_, K_.fx(), K_.fy(),
                                         K_.ox(), K_.oy(), want, timestamp_, &fresh),
                  "RgbdImagePyramid::build");
    dvo_amd_pyramid_release(handle_);
    handle_ = fresh;
    levels_ = want;
  }

  double timestamp() const { return timestamp_; }
  int width() const { return width_; }
  int height() const { return height_; }
  dvo_amd_pyramid *handle() const { return handle_; }

 private:
  int width_, height_;
  IntrinsicMatrix K_;
  int device_;
  double timestamp_;
  std::vector<float> intensity_, depth_;
  dvo_amd_pyramid *handle_;
  int levels_;
};
typedef RgbdImagePyramid::Ptr RgbdImagePyramidPtr;

// rgbd_image.h:127-144
class RgbdCameraPyramid {
 public:
  RgbdCameraPyramid(size_t base_width, size_t base_height, const IntrinsicMatrix &base_intrinsics, int device = 0)
      : width_((int)base_width), height_((int)base_height), K_(base_intrinsics), device_(device) {}

  // raw planes: intensity 0..255, depth metres with NaN = invalid, stride in floats
  RgbdImagePyramidPtr create(const float *base_intensity, const float *base_depth, int stride = 0, double timestamp = 0.0) {
    return RgbdImagePyramidPtr(new RgbdImagePyramid(width_, height_, K_, base_intensity, base_depth, stride ? stride : width_,
                                                    device_, timestamp));
  }
#ifdef DVO_AMD_HAVE_OPENCV
  // RgbdCameraPyramid::create(const cv::Mat&, const cv::Mat&), rgbd_image.h:135: CV_32FC1 planes
  RgbdImagePyramidPtr create(const cv::Mat &base_intensity, const cv::Mat &base_depth) {
    if (base_intensity.type() != CV_32FC1 || base_depth.type() != CV_32FC1 || base_intensity.size() != base_depth.size() ||
        base_intensity.cols != width_ || base_intensity.rows != height_)
      throw DvoAmdError(DVO_AMD_ERR_INVALID_ARGUMENT, "RgbdCameraPyramid::create");
    if (base_intensity.step != base_depth.step) {
      cv::Mat d = base_depth.clone(), i = base_intensity.clone();
      return create(i.ptr<float>(), d.ptr<float>(), (int)(i.step / sizeof(float)));
    }
    return create(base_intensity.ptr<float>(), base_depth.ptr<float>(), (int)(base_intensity.step / sizeof(float)));
  }
#endif
  void build(size_t) {}  // camera levels are derived together with the image levels
  const IntrinsicMatrix &intrinsics() const { return K_; }

 private:
  int width_, height_;
  IntrinsicMatrix K_;
  int device_;
};
typedef std::shared_ptr<RgbdCameraPyramid> RgbdCameraPyramidPtr;

}  // namespace core

// dense_tracking.h:39-213
class DenseTracker {
 public:
  struct Config {
    int FirstLevel, LastLevel;
    int MaxIterationsPerLevel;
    double Precision;
    double Mu;
    bool UseInitialEstimate;
    bool UseWeighting;  // never read by match() in the reference either
    float IntensityDerivativeThreshold;
    float DepthDerivativeThreshold;

    Config() {
      dvo_amd_config c;
      dvo_amd_default_config(&c);
      FirstLevel = c.first_level, LastLevel = c.last_level, MaxIterationsPerLevel = c.max_iterations_per_level;
      Precision = c.precision, Mu = c.mu, UseInitialEstimate = c.use_initial_estimate != 0, UseWeighting = true;
      IntensityDerivativeThreshold = c.intensity_derivative_threshold;
      DepthDerivativeThreshold = c.depth_derivative_threshold;
    }
    size_t getNumLevels() const { return (size_t)FirstLevel + 1; }
    bool UseEstimateSmoothing() const { return Mu > 1e-6; }
    bool IsSane() const { return FirstLevel >= LastLevel; }
  };

  struct TerminationCriteria {
    enum Enum { IterationsExceeded, IncrementTooSmall, LogLikelihoodDecreased, TooFewConstraints, NumCriteria };
  };

  struct IterationStats {
    size_t Id, ValidConstraints;
    double TDistributionLogLikelihood;
    core::Vector2d TDistributionMean;
    core::Matrix2d TDistributionPrecision;
    double PriorLogLikelihood;
    core::Vector6d EstimateIncrement;
    core::Matrix6d EstimateInformation;
  };
  typedef std::vector<IterationStats> IterationStatsVector;

  struct LevelStats {
    size_t Id, MaxValidPixels, ValidPixels;
    TerminationCriteria::Enum TerminationCriterion;
    IterationStatsVector Iterations;

    bool HasIterationWithIncrement() const {  // dense_tracking_config.cpp:129-134
      const size_t min = (TerminationCriterion == TerminationCriteria::LogLikelihoodDecreased ||
                          TerminationCriterion == TerminationCriteria::TooFewConstraints)
                             ? 2
                             : 1;
      return Iterations.size() >= min;
    }
    const IterationStats &LastIterationWithIncrement() const {
      if (!HasIterationWithIncrement()) throw std::logic_error("LevelStats: no iteration with an increment");
      return TerminationCriterion == TerminationCriteria::LogLikelihoodDecreased ? Iterations[Iterations.size() - 2]
                                                                                  : Iterations[Iterations.size() - 1];
    }
    const IterationStats &LastIteration() const { return Iterations.back(); }
  };
  typedef std::vector<LevelStats> LevelStatsVector;

  struct Stats {
    LevelStatsVector Levels;
  };

  struct Result {
    core::AffineTransformd Transformation;
    core::Matrix6d Information;
    double LogLikelihood;
    Stats Statistics;

    Result() : LogLikelihood(std::numeric_limits<double>::max()) {  // dense_tracking_config.cpp:101-108
      const double nan = std::numeric_limits<double>::quiet_NaN();
      double *t = core::data(Transformation);
      for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 3; ++r) t[c * 4 + r] = nan;
      Information.setIdentity();
    }
    bool isNaN() const {  // dense_tracking_config.cpp:96-99
      const double *t = core::data(Transformation);
      double s = 0;
      for (int i = 0; i < 16; ++i) s += t[i];
      return !std::isfinite(s) || !std::isfinite(Information.sum());
    }
    void setIdentity() {
      Transformation.setIdentity();
      Information.setIdentity();
      LogLikelihood = 0.0;
    }
    void clearStatistics() { Statistics.Levels.clear(); }
  };

  static const Config &getDefaultConfig() {
    static Config c;
    return c;
  }

  explicit DenseTracker(const Config &config = getDefaultConfig(), int device = 0) : ctx_(nullptr), device_(device) {
    dvo_amd_config c = to_c(config);
    detail::check(dvo_amd_context_create(device, &c, &ctx_), "DenseTracker::DenseTracker");
    cfg = config;
  }
  DenseTracker(const DenseTracker &other) : ctx_(nullptr), device_(other.device_) {
    dvo_amd_config c = to_c(other.cfg);
    detail::check(dvo_amd_context_create(device_, &c, &ctx_), "DenseTracker::DenseTracker");
    cfg = other.cfg;
  }
  DenseTracker &operator=(const DenseTracker &) = delete;
  ~DenseTracker() { dvo_amd_context_destroy(ctx_); }

  const Config &configuration() const { return cfg; }

  void configure(const Config &config) {
    dvo_amd_config c = to_c(config);
    detail::check(dvo_amd_configure(ctx_, &c), "DenseTracker::configure");  // assert(config.IsSane()) in the reference
    cfg = config;
  }

  // dense_tracking.cpp:99-109
  bool match(core::RgbdImagePyramid &reference, core::RgbdImagePyramid &current, core::AffineTransformd &transformation) {
    Result result;
    result.Transformation = transformation;
    const bool success = match(reference, current, result);
    transformation = result.Transformation;
    return success;
  }

  // dense_tracking.cpp:123-376
  bool match(core::RgbdImagePyramid &reference, core::RgbdImagePyramid &current, Result &result) {
    reference.compute(cfg.getNumLevels());
    current.compute(cfg.getNumLevels());
    const int cap = (cfg.FirstLevel - cfg.LastLevel + 1) * (cfg.MaxIterationsPerLevel + 1);
    std::vector<dvo_amd_iteration_stats> its((size_t)cap);
    dvo_amd_result r;
    std::memset(&r, 0, sizeof(r));
    r.iterations = its.data();
    r.iterations_capacity = cap;
    double T0[16];
    std::memcpy(T0, core::data(result.Transformation), sizeof(T0));
    detail::check(dvo_amd_match(ctx_, reference.handle(), current.handle(), cfg.UseInitialEstimate ? T0 : nullptr, &r),
                  "DenseTracker::match");
    std::memcpy(core::data(result.Transformation), r.transformation, sizeof(r.transformation));
    std::memcpy(core::data(result.Information), r.information, sizeof(r.information));
    result.LogLikelihood = r.loglik;
    for (int l = 0; l < r.n_levels; ++l) {
      LevelStats ls;
      ls.Id = (size_t)r.levels[l].id;
      ls.MaxValidPixels = (size_t)r.levels[l].max_valid_pixels;
      ls.ValidPixels = (size_t)r.levels[l].valid_pixels;
      ls.TerminationCriterion = (TerminationCriteria::Enum)r.levels[l].termination;
      for (int k = 0; k < r.levels[l].n_iterations; ++k) {
        const dvo_amd_iteration_stats &s = its[(size_t)(r.levels[l].first_iteration + k)];
        IterationStats is;
        is.Id = (size_t)s.id, is.ValidConstraints = (size_t)s.valid_constraints;
        is.TDistributionLogLikelihood = s.tdist_loglik;
        for (int i = 0; i < 2; ++i) is.TDistributionMean(i) = s.tdist_mean[i];
        for (int i = 0; i < 4; ++i) is.TDistributionPrecision(i % 2, i / 2) = s.tdist_precision[i];
        is.PriorLogLikelihood = s.prior_loglik;
        for (int i = 0; i < 6; ++i) is.EstimateIncrement(i) = s.increment[i];
        for (int i = 0; i < 36; ++i) is.EstimateInformation(i % 6, i / 6) = s.information[i];
        ls.Iterations.push_back(is);
      }
      result.Statistics.Levels.push_back(ls);
    }
    return true;  // the reference's `success` is constant true (dense_tracking.cpp:135,375)
  }

  dvo_amd_context *handle() const { return ctx_; }

 private:
  static dvo_amd_config to_c(const Config &c) {
    dvo_amd_config o;
    o.first_level = c.FirstLevel, o.last_level = c.LastLevel, o.max_iterations_per_level = c.MaxIterationsPerLevel;
    o.precision = c.Precision, o.mu = c.Mu, o.use_initial_estimate = c.UseInitialEstimate ? 1 : 0;
    o.intensity_derivative_threshold = c.IntensityDerivativeThreshold;
    o.depth_derivative_threshold = c.DepthDerivativeThreshold;
    return o;
  }

  Config cfg;
  dvo_amd_context *ctx_;
  int device_;
};

}  // namespace dvo

template <typename CharT, typename Traits>
std::basic_ostream<CharT, Traits> &operator<<(std::basic_ostream<CharT, Traits> &out, const dvo::DenseTracker::Config &c) {
  out << "First Level = " << c.FirstLevel << ", Last Level = " << c.LastLevel
      << ", Max Iterations per Level = " << c.MaxIterationsPerLevel << ", Precision = " << c.Precision << ", Mu = " << c.Mu
      << ", Use Initial Estimate = " << (c.UseInitialEstimate ? "true" : "false")
      << ", Intensity Derivative Threshold = " << c.IntensityDerivativeThreshold
      << ", Depth Derivative Threshold = " << c.DepthDerivativeThreshold;
  return out;
}

#endif  // DVO_AMD_DENSE_TRACKING_HPP_
