// dense_tracking.hpp -- header-only C++ adaptor over the C ABI (include/dvo_amd.h).
//
// Re-declares, name for name, the part of dvo_core's interface that callers of the dense-tracking hot path use:
//   dvo::core::IntrinsicMatrix        dvo_core/include/dvo/core/intrinsic_matrix.h:33-64
//   dvo::core::RgbdCameraPyramid      dvo_core/include/dvo/core/rgbd_image.h:127-144
//   dvo::core::RgbdImagePyramid       dvo_core/include/dvo/core/rgbd_image.h:242-262
//   dvo::core::RgbdImage (level(i))   dvo_core/include/dvo/core/rgbd_image.h:146-240 (what callers of the hot path touch)
//   dvo::core::PointSelection + predicates   dvo_core/include/dvo/core/point_selection.h:32-117
//   dvo::DenseTracker {Config, TerminationCriteria, IterationStats, LevelStats, Stats, Result, configure, match x4,
//                      computeIntensityErrorImage}
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
#include <iostream>
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

// weight_calculation.h:107-119,191-203.  DenseTracker::configure stores these in Config and match() never reads them (the
// bivariate t-distribution with 5 degrees of freedom is hard-coded); they exist here so that
// dvo_ros' updateConfigFromDynamicReconfigure (configtools.h:34-79) and the Config stream operator compile unchanged.
struct ScaleEstimators {
  typedef enum { Unit, NormalDistribution, TDistribution, MAD } enum_t;
  static const char *str(enum_t type) {  // weight_calculation.cpp:255-273
    switch (type) {
      case Unit: return "Unit";
      case TDistribution: return "TDistribution";
      case MAD: return "MAD";
      case NormalDistribution: return "NormalDistribution";
      default: break;
    }
    return "";
  }
};
struct InfluenceFunctions {
  typedef enum { Unit, Tukey, TDistribution, Huber } enum_t;
  static const char *str(enum_t type) {  // weight_calculation.cpp:373-391
    switch (type) {
      case Unit: return "Unit";
      case TDistribution: return "TDistribution";
      case Tukey: return "Tukey";
      case Huber: return "Huber";
      default: break;
    }
    return "";
  }
};
// TDistributionScaleEstimator::DEFAULT_DOF / TDistributionInfluenceFunction::DEFAULT_DOF (weight_calculation.cpp)
static const float kTDistributionDefaultDof = 5.0f;

namespace linalg {
// eigenvalues of a symmetric 6x6 matrix (column-major) in ascending order by cyclic Jacobi rotations: what
// Eigen::EigenSolver + std::sort give DenseTracker::IterationStats::InformationEigenValues (dense_tracking_config.cpp:123-128)
// for the symmetric EstimateInformation
inline void symmetric_eigenvalues6(const double *A_colmajor, double ev[6]) {
  double a[6][6];
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c < 6; ++c) a[r][c] = 0.5 * (A_colmajor[c * 6 + r] + A_colmajor[r * 6 + c]);
  for (int sweep = 0; sweep < 64; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < 6; ++c) (r == c ? diag : off) += a[r][c] * a[r][c];
    if (off <= 1e-30 * diag || off == 0.0) break;
    for (int p = 0; p < 5; ++p)
      for (int q = p + 1; q < 6; ++q) {
        if (a[p][q] == 0.0) continue;
        const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 6; ++k) {  // columns p, q
          const double akp = a[k][p], akq = a[k][q];
          a[k][p] = c * akp - s * akq, a[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 6; ++k) {  // rows p, q
          const double apk = a[p][k], aqk = a[q][k];
          a[p][k] = c * apk - s * aqk, a[q][k] = s * apk + c * aqk;
        }
      }
  }
  for (int i = 0; i < 6; ++i) ev[i] = a[i][i];
  std::sort(ev, ev + 6);
}
}  // namespace linalg

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
class RgbdImagePyramid;

// rgbd_image.h:146-240, as far as callers of the hot path use it: LocalTracker::update pre-builds the point cloud and the
// 8-channel acceleration structure of every level (local_tracker.cpp:163-169).  Here every level, its derivative planes,
// rays and gather layout are built on the GPU by RgbdImagePyramid::build, so these members have nothing left to do.
class RgbdImage {
 public:
  RgbdImage() : width(0), height(0), owner_(nullptr), level_(0) {}
  size_t width, height;
  IntrinsicMatrix intrinsics;  // of this level (RgbdCamera::intrinsics())
  void initialize() {}
  void calculateDerivatives() {}
  bool calculateIntensityDerivatives() { return false; }
  void calculateDepthDerivatives() {}
  void buildPointCloud() {}
  void buildAccelerationStructure() {}
  // host copy of one plane of this level: 0 intensity, 1 depth, 2 intensity_dx, 3 intensity_dy, 4 depth_dx, 5 depth_dy
  inline std::vector<float> plane(int which) const;

 private:
  friend class RgbdImagePyramid;
  const RgbdImagePyramid *owner_;
  int level_;
};

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

  // rgbd_image.h:256: builds up to idx + 1 levels on demand, like the reference
  RgbdImage &level(size_t idx) {
    if (idx >= (size_t)DVO_AMD_MAX_LEVELS) throw DvoAmdError(DVO_AMD_ERR_INVALID_ARGUMENT, "RgbdImagePyramid::level");
    build(idx + 1);
    RgbdImage &img = level_views_[idx];  // fixed storage: a reference handed out stays valid, as in the reference
    if (img.owner_ != this || img.level_ != (int)idx || img.width == 0) {
      int w = 0, h = 0;
      float k[4];
      detail::check(dvo_amd_pyramid_level_info(handle_, (int)idx, &w, &h, k), "RgbdImagePyramid::level");
      img.width = (size_t)w, img.height = (size_t)h;
      img.intrinsics = IntrinsicMatrix::create(k[0], k[1], k[2], k[3]);
      img.owner_ = this, img.level_ = (int)idx;
    }
    return img;
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
  RgbdImage level_views_[DVO_AMD_MAX_LEVELS];
};
typedef RgbdImagePyramid::Ptr RgbdImagePyramidPtr;

inline std::vector<float> RgbdImage::plane(int which) const {
  if (!owner_) throw DvoAmdError(DVO_AMD_ERR_INVALID_ARGUMENT, "RgbdImage::plane");
  std::vector<float> out(width * height);
  detail::check(dvo_amd_pyramid_download_plane(owner_->handle(), level_, which, out.data()), "RgbdImage::plane");
  return out;
}

// rgbd_image.h:39-89: the 48-byte record PointSelection::select hands out ({x, y, z, 1; I, Z, Ix, Iy, Zx, Zy, 0, 0})
struct PointWithIntensityAndDepth {
  union Point {
    float data[4];
    struct {
      float x, y, z;
    };
  };
  union IntensityAndDepth {
    float data[8];
    struct {
      float i, z, idx, idy, zdx, zdy, time_interpolation;
    };
  };
  typedef std::vector<PointWithIntensityAndDepth> VectorType;
  Point point;
  IntensityAndDepth intensity_and_depth;
};

// point_selection.h:32-67.  The selection runs on the GPU, so a predicate must be expressible as the two gradient thresholds
// of the reference's own predicates; deviceThresholds() says how.  A user-defined predicate with other logic cannot run on
// the device: PointSelection throws DvoAmdError for it instead of silently selecting something else.
class PointSelectionPredicate {
 public:
  virtual ~PointSelectionPredicate() {}
  virtual bool isPointOk(const size_t &x, const size_t &y, const float &z, const float &idx, const float &idy, const float &zdx,
                         const float &zdy) const = 0;
  virtual bool deviceThresholds(float &intensity_threshold, float &depth_threshold) const {
    (void)intensity_threshold, (void)depth_threshold;
    return false;
  }
};

class ValidPointPredicate : public PointSelectionPredicate {
 public:
  virtual ~ValidPointPredicate() {}
  virtual bool isPointOk(const size_t &, const size_t &, const float &z, const float &, const float &, const float &zdx,
                         const float &zdy) const {
    return z == z && zdx == zdx && zdy == zdy;
  }
  // |zdx| > -1 holds for every non-NaN zdx: thresholds of -1 leave exactly the validity part of the test
  virtual bool deviceThresholds(float &ti, float &td) const {
    ti = -1.0f, td = -1.0f;
    return true;
  }
};

class ValidPointAndGradientThresholdPredicate : public PointSelectionPredicate {
 public:
  float intensity_threshold;
  float depth_threshold;
  ValidPointAndGradientThresholdPredicate() : intensity_threshold(0.0f), depth_threshold(0.0f) {}
  virtual ~ValidPointAndGradientThresholdPredicate() {}
  virtual bool isPointOk(const size_t &, const size_t &, const float &z, const float &idx, const float &idy, const float &zdx,
                         const float &zdy) const {
    return z == z && zdx == zdx && zdy == zdy &&
           (std::abs(idx) > intensity_threshold || std::abs(idy) > intensity_threshold || std::abs(zdx) > depth_threshold ||
            std::abs(zdy) > depth_threshold);
  }
  virtual bool deviceThresholds(float &ti, float &td) const {
    ti = intensity_threshold, td = depth_threshold;
    return true;
  }
};

// point_selection.h:70-117.  The reference caches the compacted 48-byte records per level until setRgbdImagePyramid(); here
// the selection lives with the device pyramid (a NaN-masked depth plane per threshold pair, cached for the pyramid's
// lifetime), so this class only carries the pyramid pointer and the predicate.  select() materialises host records for
// callers that want to look at them; match() never does.
class PointSelection {
 public:
  typedef PointWithIntensityAndDepth::VectorType PointVector;
  typedef PointVector::iterator PointIterator;

  explicit PointSelection(const PointSelectionPredicate &predicate) : pyramid_(nullptr), predicate_(predicate), debug_(false) {}
  PointSelection(RgbdImagePyramid &pyramid, const PointSelectionPredicate &predicate)
      : pyramid_(&pyramid), predicate_(predicate), debug_(false) {}
  virtual ~PointSelection() {}

  RgbdImagePyramid &getRgbdImagePyramid() {
    if (!pyramid_) throw DvoAmdError(DVO_AMD_ERR_INVALID_ARGUMENT, "PointSelection::getRgbdImagePyramid");  // assert(pyramid_ != 0)
    return *pyramid_;
  }
  void setRgbdImagePyramid(RgbdImagePyramid &pyramid) {
    pyramid_ = &pyramid;
    storage_.clear();  // point_selection.cpp:51-59: the cache belongs to the previous pyramid
  }
  void recycle(RgbdImagePyramid &pyramid) { setRgbdImagePyramid(pyramid); }

  // point_selection.cpp:68-71
  size_t getMaximumNumberOfPoints(const size_t &level) {
    RgbdImagePyramid &p = getRgbdImagePyramid();
    return size_t((double)((size_t)p.width() * (size_t)p.height()) * std::pow(0.25, double(level)));
  }

  // the thresholds the device selection runs with (throws for a predicate the device cannot evaluate)
  void thresholds(float &ti, float &td) const {
    if (!predicate_.deviceThresholds(ti, td))
      throw DvoAmdError(DVO_AMD_ERR_INVALID_ARGUMENT, "PointSelection: predicate has no device form (deviceThresholds)");
  }

  // number of selected pixels of a level (the distance last_point - first_point of select())
  size_t size(const size_t &level) {
    RgbdImagePyramid &p = getRgbdImagePyramid();
    p.build(level + 1);
    float ti, td;
    thresholds(ti, td);
    int count = 0;
    detail::check(dvo_amd_pyramid_select(p.handle(), (int)level, ti, td, &count, nullptr), "PointSelection::size");
    return (size_t)count;
  }

  // point_selection.cpp:89-152: row-major scan, predicate, 48-byte records.  Host copy, built on demand and cached per level.
  void select(const size_t &level, PointIterator &first_point, PointIterator &last_point) {
    RgbdImagePyramid &p = getRgbdImagePyramid();
    RgbdImage &img = p.level(level);
    if (storage_.size() < level + 1) storage_.resize(level + 1);
    Storage &st = storage_[level];
    if (!st.is_cached || debug_) {
      float ti, td;
      thresholds(ti, td);
      const size_t n = img.width * img.height;
      std::vector<unsigned char> mask(n);
      int count = 0;
      detail::check(dvo_amd_pyramid_select(p.handle(), (int)level, ti, td, &count, mask.data()), "PointSelection::select");
      std::vector<float> pl[6];
      for (int k = 0; k < 6; ++k) pl[k] = img.plane(k);
      st.points.assign((size_t)count, PointWithIntensityAndDepth());
      const float fx = img.intrinsics.fx(), fy = img.intrinsics.fy(), ox = img.intrinsics.ox(), oy = img.intrinsics.oy();
      size_t o = 0;
      for (size_t y = 0; y < img.height; ++y)
        for (size_t x = 0; x < img.width; ++x) {
          const size_t i = y * img.width + x;
          if (!mask[i]) continue;
          PointWithIntensityAndDepth &q = st.points[o++];
          const float z = pl[1][i];
          // RgbdCamera's ray template ((x - ox) / fx, (y - oy) / fy, 1, 0) times depth, w = 1 (rgbd_image.cpp:198-199,245-262)
          q.point.data[0] = (((float)x - ox) / fx) * z, q.point.data[1] = (((float)y - oy) / fy) * z;
          q.point.data[2] = z, q.point.data[3] = 1.0f;
          for (int k = 0; k < 6; ++k) q.intensity_and_depth.data[k] = pl[k][i];
          q.intensity_and_depth.data[6] = q.intensity_and_depth.data[7] = 0.0f;
        }
      if (debug_) st.debug_idx.swap(mask);
      st.is_cached = true;
    }
    first_point = st.points.begin();
    last_point = st.points.end();
  }

  // point_selection.cpp:74-86 (a row-major 0/1 byte per pixel instead of a cv::Mat)
  bool getDebugIndex(const size_t &level, std::vector<unsigned char> &dbg_idx) {
    if (debug_ && storage_.size() > level) {
      dbg_idx = storage_[level].debug_idx;
      return !dbg_idx.empty();
    }
    return false;
  }
  void debug(bool v) { debug_ = v; }
  bool debug() const { return debug_; }

 private:
  struct Storage {
    PointVector points;
    bool is_cached;
    std::vector<unsigned char> debug_idx;
    Storage() : is_cached(false) {}
  };
  RgbdImagePyramid *pyramid_;
  std::vector<Storage> storage_;
  const PointSelectionPredicate &predicate_;
  bool debug_;
};

// surface_pyramid.h:34-49 / surface_pyramid.cpp:44-105: uint16 raw depth -> float metres, 0 -> NaN, one fp32 multiply per pixel.
// Host-side shim for callers that build the float depth image themselves (camera_dense_tracking.cpp:233-236); the same
// conversion runs on the GPU when a pyramid is created straight from the raw frame (dvo_amd_pyramid_create_raw).  Both reference entry points give the same values; the Sse one needs 16-byte aligned rows of a
// multiple of 8 pixels, which this shim does not.
class SurfacePyramid {
 public:
  static void convertRawDepthImage(const unsigned short *input, float *output, size_t n_pixels, float scale) {
    const float nan = std::numeric_limits<float>::quiet_NaN();
    for (size_t i = 0; i < n_pixels; ++i) output[i] = input[i] == 0 ? nan : (float)input[i] * scale;
  }
  static void convertRawDepthImageSse(const unsigned short *input, float *output, size_t n_pixels, float scale) {
    convertRawDepthImage(input, output, n_pixels, scale);
  }
#ifdef DVO_AMD_HAVE_OPENCV
  static void convertRawDepthImage(const cv::Mat &input, cv::Mat &output, float scale) {
    output.create(input.rows, input.cols, CV_32FC1);
    for (int y = 0; y < input.rows; ++y) convertRawDepthImage(input.ptr<unsigned short>(y), output.ptr<float>(y), (size_t)input.cols, scale);
  }
  static void convertRawDepthImageSse(const cv::Mat &input, cv::Mat &output, float scale) { convertRawDepthImage(input, output, scale); }
#endif
  SurfacePyramid() {}
  virtual ~SurfacePyramid() {}
};

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
    // The next six fields are declared by the reference (dense_tracking.h:51-60), assigned by its callers (configtools.h:70-79)
    // and never read by match(): kept so that those call sites compile; they do not cross the C ABI.
    bool UseWeighting;
    bool UseParallel;  // (not even initialised by the reference's constructor; false here)
    core::InfluenceFunctions::enum_t InfluenceFuntionType;  // [sic]
    float InfluenceFunctionParam;
    core::ScaleEstimators::enum_t ScaleEstimatorType;
    float ScaleEstimatorParam;
    float IntensityDerivativeThreshold;
    float DepthDerivativeThreshold;
    // NOT a field of the reference: dvo_amd_config::segment_geometry (DVO_AMD_GEOMETRY_THROUGHPUT, the default, or
    // DVO_AMD_GEOMETRY_LATENCY for the shortest single match()); callers of the reference never touch it
    int SegmentGeometry;

    Config() {  // dense_tracking_config.cpp:27-42
      dvo_amd_config c;
      dvo_amd_default_config(&c);
      FirstLevel = c.first_level, LastLevel = c.last_level, MaxIterationsPerLevel = c.max_iterations_per_level;
      Precision = c.precision, Mu = c.mu, UseInitialEstimate = c.use_initial_estimate != 0, UseWeighting = true;
      UseParallel = false;
      InfluenceFuntionType = core::InfluenceFunctions::TDistribution, InfluenceFunctionParam = core::kTDistributionDefaultDof;
      ScaleEstimatorType = core::ScaleEstimators::TDistribution, ScaleEstimatorParam = core::kTDistributionDefaultDof;
      IntensityDerivativeThreshold = c.intensity_derivative_threshold;
      DepthDerivativeThreshold = c.depth_derivative_threshold;
      SegmentGeometry = c.segment_geometry;
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

    // dense_tracking_config.cpp:123-136 (caller: keyframe_graph.cpp:370-371)
    void InformationEigenValues(core::Vector6d &eigenvalues) const {
      double ev[6];
      core::linalg::symmetric_eigenvalues6(EstimateInformation.data(), ev);
      for (int i = 0; i < 6; ++i) eigenvalues(i) = ev[i];
    }
    double InformationConditionNumber() const {
      core::Vector6d ev;
      InformationEigenValues(ev);
      return std::abs(ev(5) / ev(0));
    }
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
    // dense_tracking_config.cpp:138-172 (the reference prints "awkward" + the level and asserts; no assert crosses this adaptor)
    const IterationStats &LastIterationWithIncrement() const {
      if (!HasIterationWithIncrement()) throw std::logic_error("LevelStats: no iteration with an increment");
      return TerminationCriterion == TerminationCriteria::LogLikelihoodDecreased ? Iterations[Iterations.size() - 2]
                                                                                  : Iterations[Iterations.size() - 1];
    }
    IterationStats &LastIterationWithIncrement() {
      return const_cast<IterationStats &>(static_cast<const LevelStats *>(this)->LastIterationWithIncrement());
    }
    const IterationStats &LastIteration() const { return Iterations.back(); }
    IterationStats &LastIteration() { return Iterations.back(); }
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

  explicit DenseTracker(const Config &config = getDefaultConfig(), int device = 0)
      : ctx_(nullptr), device_(device), reference_selection_(selection_predicate_) {
    dvo_amd_config c = to_c(config);
    detail::check(dvo_amd_context_create(device, &c, &ctx_), "DenseTracker::DenseTracker");
    cfg = config;
    selection_predicate_.intensity_threshold = cfg.IntensityDerivativeThreshold;
    selection_predicate_.depth_threshold = cfg.DepthDerivativeThreshold;
  }
  DenseTracker(const DenseTracker &other) : ctx_(nullptr), device_(other.device_), reference_selection_(selection_predicate_) {
    dvo_amd_config c = to_c(other.cfg);
    detail::check(dvo_amd_context_create(device_, &c, &ctx_), "DenseTracker::DenseTracker");
    cfg = other.cfg;
    selection_predicate_.intensity_threshold = cfg.IntensityDerivativeThreshold;
    selection_predicate_.depth_threshold = cfg.DepthDerivativeThreshold;
  }
  DenseTracker &operator=(const DenseTracker &) = delete;
  ~DenseTracker() { dvo_amd_context_destroy(ctx_); }

  const Config &configuration() const { return cfg; }

  void configure(const Config &config) {
    dvo_amd_config c = to_c(config);
    detail::check(dvo_amd_configure(ctx_, &c), "DenseTracker::configure");  // assert(config.IsSane()) in the reference
    cfg = config;
    selection_predicate_.intensity_threshold = cfg.IntensityDerivativeThreshold;  // dense_tracking.cpp:76-77
    selection_predicate_.depth_threshold = cfg.DepthDerivativeThreshold;
  }

  // dense_tracking.cpp:99-109
  bool match(core::RgbdImagePyramid &reference, core::RgbdImagePyramid &current, core::AffineTransformd &transformation) {
    Result result;
    result.Transformation = transformation;
    const bool success = match(reference, current, result);
    transformation = result.Transformation;
    return success;
  }

  // dense_tracking.cpp:111-121
  bool match(core::PointSelection &reference, core::RgbdImagePyramid &current, core::AffineTransformd &transformation) {
    Result result;
    result.Transformation = transformation;
    const bool success = match(reference, current, result);
    transformation = result.Transformation;
    return success;
  }

  // dense_tracking.cpp:123-129
  bool match(core::RgbdImagePyramid &reference, core::RgbdImagePyramid &current, Result &result) {
    reference.compute(cfg.getNumLevels());
    reference_selection_.setRgbdImagePyramid(reference);
    return match(reference_selection_, current, result);
  }

  // dense_tracking.cpp:131-376: the reference pixels are the ones `reference`'s predicate keeps
  bool match(core::PointSelection &reference, core::RgbdImagePyramid &current, Result &result) {
    core::RgbdImagePyramid &ref_pyramid = reference.getRgbdImagePyramid();
    ref_pyramid.compute(cfg.getNumLevels());
    current.compute(cfg.getNumLevels());
    float ti = 0.0f, td = 0.0f;
    reference.thresholds(ti, td);
    const int cap = (cfg.FirstLevel - cfg.LastLevel + 1) * (cfg.MaxIterationsPerLevel + 1);
    std::vector<dvo_amd_iteration_stats> its((size_t)cap);
    dvo_amd_result r;
    std::memset(&r, 0, sizeof(r));
    r.iterations = its.data();
    r.iterations_capacity = cap;
    double T0[16];
    std::memcpy(T0, core::data(result.Transformation), sizeof(T0));
    detail::check(dvo_amd_match_selection(ctx_, ref_pyramid.handle(), ti, td, current.handle(),
                                          cfg.UseInitialEstimate ? T0 : nullptr, &r),
                  "DenseTracker::match");
    std::memcpy(core::data(result.Transformation), r.transformation, sizeof(r.transformation));
    std::memcpy(core::data(result.Information), r.information, sizeof(r.information));
    result.LogLikelihood = r.loglik;
    result.Statistics.Levels.clear();
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

  // dense_tracking.cpp:378-444 (caller: keyframe_graph.cpp:354): |intensity residual| per reference pixel of `level`, 0 where
  // the pixel is not selected or its warp is invalid; CV_32FC1 where OpenCV is available, else ErrorImage
  struct ErrorImage {
    int rows, cols;
    std::vector<float> data;
    float &at(int y, int x) { return data[(size_t)y * cols + x]; }
    float at(int y, int x) const { return data[(size_t)y * cols + x]; }
  };
  ErrorImage computeIntensityErrorImageRaw(core::RgbdImagePyramid &reference, core::RgbdImagePyramid &current,
                                           const core::AffineTransformd &transformation, size_t level = 0) {
    reference.compute(level + 1);
    current.compute(level + 1);
    core::RgbdImage &img = reference.level(level);
    ErrorImage out;
    out.rows = (int)img.height, out.cols = (int)img.width;
    out.data.resize(img.width * img.height);
    detail::check(dvo_amd_error_image(ctx_, reference.handle(), current.handle(), core::data(transformation), (int)level,
                                      out.data.data()),
                  "DenseTracker::computeIntensityErrorImage");
    return out;
  }
#ifdef DVO_AMD_HAVE_OPENCV
  cv::Mat computeIntensityErrorImage(core::RgbdImagePyramid &reference, core::RgbdImagePyramid &current,
                                     const core::AffineTransformd &transformation, size_t level = 0) {
    ErrorImage e = computeIntensityErrorImageRaw(reference, current, transformation, level);
    cv::Mat m(e.rows, e.cols, CV_32FC1);
    for (int y = 0; y < e.rows; ++y) std::memcpy(m.ptr<float>(y), &e.data[(size_t)y * e.cols], sizeof(float) * (size_t)e.cols);
    return m;
  }
#else
  ErrorImage computeIntensityErrorImage(core::RgbdImagePyramid &reference, core::RgbdImagePyramid &current,
                                        const core::AffineTransformd &transformation, size_t level = 0) {
    return computeIntensityErrorImageRaw(reference, current, transformation, level);
  }
#endif

  dvo_amd_context *handle() const { return ctx_; }

 private:
  static dvo_amd_config to_c(const Config &c) {
    dvo_amd_config o;
    o.first_level = c.FirstLevel, o.last_level = c.LastLevel, o.max_iterations_per_level = c.MaxIterationsPerLevel;
    o.precision = c.Precision, o.mu = c.Mu, o.use_initial_estimate = c.UseInitialEstimate ? 1 : 0;
    o.intensity_derivative_threshold = c.IntensityDerivativeThreshold;
    o.depth_derivative_threshold = c.DepthDerivativeThreshold;
    o.segment_geometry = c.SegmentGeometry, o.reserved = 0;
    return o;
  }

  Config cfg;
  dvo_amd_context *ctx_;
  int device_;
  core::ValidPointAndGradientThresholdPredicate selection_predicate_;  // dense_tracking.h:199-200
  core::PointSelection reference_selection_;
};

}  // namespace dvo

// dense_tracking.h:215-237
template <typename CharT, typename Traits>
std::ostream &operator<<(std::basic_ostream<CharT, Traits> &out, const dvo::DenseTracker::Config &config) {
  out << "First Level = " << config.FirstLevel << ", Last Level = " << config.LastLevel
      << ", Max Iterations per Level = " << config.MaxIterationsPerLevel << ", Precision = " << config.Precision
      << ", Mu = " << config.Mu << ", Use Initial Estimate = " << (config.UseInitialEstimate ? "true" : "false")
      << ", Use Weighting = " << (config.UseWeighting ? "true" : "false")
      << ", Scale Estimator = " << dvo::core::ScaleEstimators::str(config.ScaleEstimatorType)
      << ", Scale Estimator Param = " << config.ScaleEstimatorParam
      << ", Influence Function = " << dvo::core::InfluenceFunctions::str(config.InfluenceFuntionType)
      << ", Influence Function Param = " << config.InfluenceFunctionParam
      << ", Intensity Derivative Threshold = " << config.IntensityDerivativeThreshold
      << ", Depth Derivative Threshold = " << config.DepthDerivativeThreshold;
  return out;
}

// dense_tracking.h:239-245
template <typename CharT, typename Traits>
std::ostream &operator<<(std::basic_ostream<CharT, Traits> &o, const dvo::DenseTracker::IterationStats &s) {
  o << "Iteration: " << s.Id << " ValidConstraints: " << s.ValidConstraints << " DataLogLikelihood: " << s.TDistributionLogLikelihood
    << " PriorLogLikelihood: " << s.PriorLogLikelihood << std::endl;
  return o;
}

// dense_tracking.h:247-279
template <typename CharT, typename Traits>
std::ostream &operator<<(std::basic_ostream<CharT, Traits> &o, const dvo::DenseTracker::LevelStats &s) {
  std::string termination;
  switch (s.TerminationCriterion) {
    case dvo::DenseTracker::TerminationCriteria::IterationsExceeded: termination = "IterationsExceeded"; break;
    case dvo::DenseTracker::TerminationCriteria::IncrementTooSmall: termination = "IncrementTooSmall"; break;
    case dvo::DenseTracker::TerminationCriteria::LogLikelihoodDecreased: termination = "LogLikelihoodDecreased"; break;
    case dvo::DenseTracker::TerminationCriteria::TooFewConstraints: termination = "TooFewConstraints"; break;
    default: break;
  }
  o << "Level: " << s.Id << " Pixel: " << s.ValidPixels << "/" << s.MaxValidPixels << " Termination: " << termination
    << " Iterations: " << s.Iterations.size() << std::endl;
  for (dvo::DenseTracker::IterationStatsVector::const_iterator it = s.Iterations.begin(); it != s.Iterations.end(); ++it) o << *it;
  return o;
}

// dense_tracking.h:281-291
template <typename CharT, typename Traits>
std::ostream &operator<<(std::basic_ostream<CharT, Traits> &o, const dvo::DenseTracker::Stats &s) {
  o << s.Levels.size() << " levels" << std::endl;
  for (dvo::DenseTracker::LevelStatsVector::const_iterator it = s.Levels.begin(); it != s.Levels.end(); ++it) o << *it;
  return o;
}

#endif  // DVO_AMD_DENSE_TRACKING_HPP_
