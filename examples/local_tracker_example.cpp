// The dense-tracking part of dvo_slam's front end, written against the forwarding headers (include/dvo_amd_compat) exactly as
// dvo_slam/src/local_tracker.cpp:40-74,127-213 writes it against dvo_core: two DenseTracker instances, two PointSelections
// sharing one predicate, keyframe -> frame and last frame -> frame matched side by side, the keyframe's selection swapped
// when a new local map starts.  Stand-ins only for what is not dense tracking: tbb::parallel_invoke -> two std::threads,
// LocalMap / g2o -> a struct that remembers frames and poses, boost::signals2 accept callbacks -> one function.
// Reads N >= 3 frames (float32 intensity / depth planes) written by tests/test_cpp_adaptor.py and prints one line per update.
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include <dvo/core/point_selection.h>
#include <dvo/core/point_selection_predicates.h>  // local_tracker.cpp:24
#include <dvo/core/rgbd_image.h>
#include <dvo/dense_tracking.h>

namespace dvo_slam {

typedef std::shared_ptr<dvo::DenseTracker> DenseTrackerPtr;
typedef std::shared_ptr<dvo::core::PointSelection> PointSelectionPtr;

// column-major 4x4 helpers standing in for Eigen::Affine3d arithmetic (only used outside the tracker)
static dvo::core::AffineTransformd inverse(const dvo::core::AffineTransformd &T) {
  dvo::core::AffineTransformd R;
  const double *t = dvo::core::data(T);
  double *r = dvo::core::data(R);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r[j * 4 + i] = t[i * 4 + j];
  for (int i = 0; i < 3; ++i) r[12 + i] = -(r[i] * t[12] + r[4 + i] * t[13] + r[8 + i] * t[14]);
  r[3] = r[7] = r[11] = 0.0, r[15] = 1.0;
  return R;
}

struct LocalMap {  // stand-in for dvo_slam::LocalMap (local_map.cpp): the frames and measurements the tracker hands over
  dvo::core::RgbdImagePyramid::Ptr keyframe, current;
  int frames, keyframe_measurements, odometry_measurements;
};

class LocalTracker {
 public:
  typedef dvo::DenseTracker::Result TrackingResult;

  // local_tracker.cpp:59-68
  LocalTracker() : force_(false), new_maps(0) {
    keyframe_tracker_.reset(new dvo::DenseTracker());
    odometry_tracker_.reset(new dvo::DenseTracker());
    last_keyframe_pose_.setIdentity();
    keyframe_points_.reset(new dvo::core::PointSelection(predicate));
    active_frame_points_.reset(new dvo::core::PointSelection(predicate));
  }

  // local_tracker.cpp:107-125
  void configure(const dvo::DenseTracker::Config &config) {
    keyframe_tracker_->configure(config);
    odometry_tracker_->configure(config);
    if (predicate.intensity_threshold != config.IntensityDerivativeThreshold ||
        predicate.depth_threshold != config.DepthDerivativeThreshold) {
      predicate.intensity_threshold = config.IntensityDerivativeThreshold;
      predicate.depth_threshold = config.DepthDerivativeThreshold;
      if (local_map_) keyframe_points_->setRgbdImagePyramid(*local_map_->keyframe);
    }
  }

  // local_tracker.cpp:127-140
  void initNewLocalMap(const dvo::core::RgbdImagePyramid::Ptr &keyframe, const dvo::core::RgbdImagePyramid::Ptr &frame) {
    keyframe_points_->setRgbdImagePyramid(*keyframe);
    active_frame_points_->setRgbdImagePyramid(*frame);
    TrackingResult r_odometry;
    r_odometry.Transformation.setIdentity();
    odometry_tracker_->match(*keyframe_points_, *frame, r_odometry);
    last_keyframe_pose_ = r_odometry.Transformation;
    initNewLocalMap(keyframe, frame, r_odometry);
  }

  // local_tracker.cpp:142-155
  void initNewLocalMap(const dvo::core::RgbdImagePyramid::Ptr &keyframe, const dvo::core::RgbdImagePyramid::Ptr &frame,
                       TrackingResult &r_odometry) {
    if (r_odometry.isNaN()) r_odometry.setIdentity();
    local_map_.reset(new LocalMap());
    local_map_->keyframe = keyframe, local_map_->current = frame;
    local_map_->frames = 2, local_map_->keyframe_measurements = 1, local_map_->odometry_measurements = 0;
    ++new_maps;
  }

  // local_tracker.cpp:157-213
  void update(const dvo::core::RgbdImagePyramid::Ptr &image, TrackingResult &r_odometry, TrackingResult &r_keyframe) {
    const dvo::DenseTracker::Config &config = keyframe_tracker_->configuration();
    image->build(config.getNumLevels());
    for (int idx = config.LastLevel; idx <= config.FirstLevel; ++idx) {
      image->level(idx).buildPointCloud();
      image->level(idx).buildAccelerationStructure();
    }
    r_odometry.Transformation.setIdentity();
    r_keyframe.Transformation = inverse(last_keyframe_pose_);

    // recycle, so we can reuse the allocated memory
    active_frame_points_->setRgbdImagePyramid(*local_map_->current);

    std::thread h1(&LocalTracker::match, keyframe_tracker_, keyframe_points_, image, &r_keyframe);  // tbb::parallel_invoke(h1, h2)
    std::thread h2(&LocalTracker::match, odometry_tracker_, active_frame_points_, image, &r_odometry);
    h1.join();
    h2.join();

    force_ = force_ || r_odometry.isNaN() || r_keyframe.isNaN();
    if (accept(r_odometry, r_keyframe) && !force_) {
      local_map_->current = image;
      local_map_->frames++, local_map_->odometry_measurements++, local_map_->keyframe_measurements++;
      last_keyframe_pose_ = r_keyframe.Transformation;
    } else {
      force_ = false;
      keyframe_points_.swap(active_frame_points_);
      dvo::core::RgbdImagePyramid::Ptr old_current = local_map_->current;
      initNewLocalMap(old_current, image, r_odometry);
      last_keyframe_pose_ = r_odometry.Transformation;
    }
  }

  // stand-in for the accept callbacks of KeyframeTracker (keyframe_tracker.cpp:105-190): keep the keyframe while the camera
  // stays within max_translation of it and enough of the selected pixels are still valid constraints
  bool accept(const TrackingResult &, const TrackingResult &r_keyframe) const {
    const double *t = dvo::core::data(r_keyframe.Transformation);
    const double d2 = t[12] * t[12] + t[13] * t[13] + t[14] * t[14];
    const dvo::DenseTracker::LevelStats &l = r_keyframe.Statistics.Levels.back();
    const double ratio = (double)l.Iterations.back().ValidConstraints / (double)l.ValidPixels;  // keyframe_tracker.cpp:167
    return d2 <= max_translation * max_translation && ratio >= 0.3;
  }

  int new_maps_started() const { return new_maps; }
  double max_translation = 0.02;

 private:
  static void match(const DenseTrackerPtr &tracker, const PointSelectionPtr &ref, const dvo::core::RgbdImagePyramid::Ptr &cur,
                    TrackingResult *r) {
    tracker->match(*ref, *cur, *r);  // local_tracker.cpp:51-54
  }

  DenseTrackerPtr keyframe_tracker_, odometry_tracker_;
  dvo::core::ValidPointAndGradientThresholdPredicate predicate;
  dvo::core::AffineTransformd last_keyframe_pose_;
  PointSelectionPtr keyframe_points_, active_frame_points_;
  bool force_;
  std::shared_ptr<LocalMap> local_map_;
  int new_maps;
};

}  // namespace dvo_slam

static std::vector<float> read_plane(const std::string &path, size_t n) {
  std::vector<float> v(n);
  FILE *f = std::fopen(path.c_str(), "rb");
  if (!f || std::fread(v.data(), sizeof(float), n, f) != n) {
    std::fprintf(stderr, "cannot read %s\n", path.c_str());
    std::exit(2);
  }
  std::fclose(f);
  return v;
}

static void print_pose(const char *tag, int frame, const dvo::DenseTracker::Result &r) {
  const double *T = dvo::core::data(r.Transformation);
  std::printf("%s %d nan %d", tag, frame, r.isNaN() ? 1 : 0);
  for (int i = 0; i < 16; ++i) std::printf(" %.17g", T[i]);
  std::printf("\n");
}

int main(int argc, char **argv) {
  if (argc < 11) {
    std::fprintf(stderr, "usage: %s dir n_frames w h fx fy ox oy intensity_threshold depth_threshold\n", argv[0]);
    return 2;
  }
  const std::string dir = argv[1];
  const int n_frames = std::atoi(argv[2]), w = std::atoi(argv[3]), h = std::atoi(argv[4]);
  const size_t n = (size_t)w * h;
  dvo::core::IntrinsicMatrix K = dvo::core::IntrinsicMatrix::create((float)std::atof(argv[5]), (float)std::atof(argv[6]),
                                                                    (float)std::atof(argv[7]), (float)std::atof(argv[8]));
  dvo::core::RgbdCameraPyramid camera(w, h, K);
  std::vector<dvo::core::RgbdImagePyramid::Ptr> frames;
  for (int i = 0; i < n_frames; ++i) {
    std::vector<float> I = read_plane(dir + "/" + std::to_string(i) + "_i.f32", n), Z = read_plane(dir + "/" + std::to_string(i) + "_z.f32", n);
    frames.push_back(camera.create(I.data(), Z.data(), 0, 0.1 * i));
  }
  dvo_slam::LocalTracker tracker;
  dvo::DenseTracker::Config cfg = dvo::DenseTracker::getDefaultConfig();  // levels 3..1, like dvo_slam
  cfg.UseInitialEstimate = true;
  cfg.IntensityDerivativeThreshold = (float)std::atof(argv[9]);
  cfg.DepthDerivativeThreshold = (float)std::atof(argv[10]);
  tracker.configure(cfg);
  tracker.initNewLocalMap(frames[0], frames[1]);
  for (int i = 2; i < n_frames; ++i) {
    dvo_slam::LocalTracker::TrackingResult r_odometry, r_keyframe;
    const int maps_before = tracker.new_maps_started();
    tracker.update(frames[(size_t)i], r_odometry, r_keyframe);
    print_pose("odometry", i, r_odometry);
    print_pose("keyframe", i, r_keyframe);
    std::printf("newmap %d %d\n", i, tracker.new_maps_started() - maps_before);
  }
  // the other two members of the boundary local callers use: the selection's size / records and the debug error image
  dvo::core::ValidPointAndGradientThresholdPredicate pred;
  pred.intensity_threshold = cfg.IntensityDerivativeThreshold, pred.depth_threshold = cfg.DepthDerivativeThreshold;
  dvo::core::PointSelection sel(*frames[0], pred);
  dvo::core::PointSelection::PointIterator first, last;
  sel.select(1, first, last);
  double sz = 0.0;
  for (dvo::core::PointSelection::PointIterator it = first; it != last; ++it) sz += it->point.z;
  std::printf("selection level 1 count %zu max %zu size %zu zsum %.9g\n", (size_t)(last - first), sel.getMaximumNumberOfPoints(1),
              sel.size(1), sz);
  dvo::DenseTracker dbg(cfg);
  dvo::core::AffineTransformd eye;
  dvo::DenseTracker::ErrorImage err = dbg.computeIntensityErrorImageRaw(*frames[0], *frames[1], eye, 1);
  double esum = 0.0;
  for (size_t i = 0; i < err.data.size(); ++i) esum += err.data[i];
  std::printf("errorimage %d %d %.9g\n", err.rows, err.cols, esum);
  return 0;
}
