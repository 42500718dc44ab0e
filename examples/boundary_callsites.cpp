// Compile (and run) check of the reference call sites that round 2's adaptor still broke -- the same members, enums and
// conversions those call sites use, restated in this repository's own words (round 4: no reference text in here) -- against the
// forwarding headers (include/dvo_amd_compat):
//   dvo_ros/include/dvo_ros/util/configtools.h:32-82     updateConfigFromDynamicReconfigure: assigns EVERY Config field,
//                                                         including the five match() never reads
//   dvo_slam/src/keyframe_graph.cpp:354-371               Statistics streamed to std::cerr, LastIterationWithIncrement()
//                                                         .InformationConditionNumber() of the finest and coarsest level
//   dvo_ros/src/camera_dense_tracking.cpp:27,233-236      #include <dvo/core/surface_pyramid.h>, convertRawDepthImageSse
//   dvo_slam/src/local_tracker.cpp:24                     #include <dvo/core/point_selection_predicates.h>
// Stand-ins only for what is not dvo_core: the dynamic_reconfigure struct dvo_ros generates from cfg/dvo.cfg and cv::Mat.
// Needs no GPU: nothing here calls a compute entry point (tests/test_cpp_adaptor.py runs it on the CPU).
#include <cassert>
#include <cmath>
#include <cstdio>
#include <iostream>
#include <sstream>
#include <vector>

#include <dvo/core/point_selection_predicates.h>
#include <dvo/core/surface_pyramid.h>
#include <dvo/core/weight_calculation.h>
#include <dvo/dense_tracking.h>

namespace dvo_ros {
// what dynamic_reconfigure generates from dvo_ros/cfg/dvo.cfg:42-57 (field names and enum constants as in the .cfg)
enum {
  CameraDenseTracker_NormalDistributionScaleEstimator = 1,
  CameraDenseTracker_TDistributionScaleEstimator = 2,
  CameraDenseTracker_MADScaleEstimator = 3,
  CameraDenseTracker_TukeyInfluenceFunction = 1,
  CameraDenseTracker_TDistributionInfluenceFunction = 2,
  CameraDenseTracker_HuberInfluenceFunction = 3
};
struct CameraDenseTrackerConfig {
  int coarsest_level, finest_level, max_iterations;
  double precision, mu;
  bool use_initial_estimate, use_weighting;
  int scale_estimator, influence_function;
  double scale_estimator_param, influence_function_param;
  double min_intensity_deriv, min_depth_deriv;
};

namespace util {

// What configtools.h:32-82 does, restated (not the reference's text): the two dynamic_reconfigure enumerators are mapped to the
// dvo_core enums, then EVERY field of DenseTracker::Config is assigned -- the five match() never reads included.  What is under
// test is that each of these member names, enum names and implicit conversions exists in the adaptor.
inline dvo::core::ScaleEstimators::enum_t scaleEstimatorOf(int v)
{
  static const dvo::core::ScaleEstimators::enum_t by_cfg_value[] = {
      dvo::core::ScaleEstimators::Unit,  // (0 is not a value of the .cfg enum)
      dvo::core::ScaleEstimators::NormalDistribution, dvo::core::ScaleEstimators::TDistribution, dvo::core::ScaleEstimators::MAD};
  assert(v >= dvo_ros::CameraDenseTracker_NormalDistributionScaleEstimator && v <= dvo_ros::CameraDenseTracker_MADScaleEstimator);
  return by_cfg_value[v];
}
inline dvo::core::InfluenceFunctions::enum_t influenceFunctionOf(int v)
{
  static const dvo::core::InfluenceFunctions::enum_t by_cfg_value[] = {
      dvo::core::InfluenceFunctions::Unit,  // (0 is not a value of the .cfg enum)
      dvo::core::InfluenceFunctions::Tukey, dvo::core::InfluenceFunctions::TDistribution, dvo::core::InfluenceFunctions::Huber};
  assert(v >= dvo_ros::CameraDenseTracker_TukeyInfluenceFunction && v <= dvo_ros::CameraDenseTracker_HuberInfluenceFunction);
  return by_cfg_value[v];
}
void updateConfigFromDynamicReconfigure(const dvo_ros::CameraDenseTrackerConfig& in, dvo::DenseTracker::Config& out)
{
  out.ScaleEstimatorType = scaleEstimatorOf(in.scale_estimator), out.ScaleEstimatorParam = in.scale_estimator_param;
  out.InfluenceFuntionType = influenceFunctionOf(in.influence_function), out.InfluenceFunctionParam = in.influence_function_param;
  out.FirstLevel = in.coarsest_level, out.LastLevel = in.finest_level;
  out.MaxIterationsPerLevel = in.max_iterations, out.Precision = in.precision, out.Mu = in.mu;
  out.UseInitialEstimate = in.use_initial_estimate, out.UseWeighting = in.use_weighting;
  out.IntensityDerivativeThreshold = in.min_intensity_deriv, out.DepthDerivativeThreshold = in.min_depth_deriv;
}

} /* namespace util */
} /* namespace dvo_ros */

namespace dvo_slam {
namespace LocalTracker { typedef dvo::DenseTracker::Result TrackingResult; }

// The members keyframe_graph.cpp:364-371 touches on a stored tracking result (restated): the whole Statistics through
// operator<<, and the information condition number of the last accepted iteration on the finest and on the coarsest level.
void printConstraintStatistics(LocalTracker::TrackingResult& result)
{
  dvo::DenseTracker::LevelStats &finest = result.Statistics.Levels.back(), &coarsest = result.Statistics.Levels.front();  // (non-const: so is LastIterationWithIncrement())
  std::cerr << result.Statistics << "\n"
            << "condition number, finest level: " << finest.LastIterationWithIncrement().InformationConditionNumber() << "\n"
            << "condition number, coarsest level: " << coarsest.LastIterationWithIncrement().InformationConditionNumber() << std::endl;
}
}  // namespace dvo_slam

int main() {
  // ---- configtools.h
  dvo_ros::CameraDenseTrackerConfig rc = {3, 1, 50, 5e-7, 0.05, true, true, dvo_ros::CameraDenseTracker_MADScaleEstimator,
                                          dvo_ros::CameraDenseTracker_HuberInfluenceFunction, 4.5, 1.345, 0.01, 0.02};
  dvo::DenseTracker::Config cfg;
  assert(cfg.InfluenceFuntionType == dvo::core::InfluenceFunctions::TDistribution && cfg.ScaleEstimatorParam == 5.0f);  // defaults
  dvo_ros::util::updateConfigFromDynamicReconfigure(rc, cfg);
  assert(cfg.FirstLevel == 3 && cfg.LastLevel == 1 && cfg.MaxIterationsPerLevel == 50 && cfg.UseInitialEstimate && cfg.Mu == 0.05);
  assert(cfg.ScaleEstimatorType == dvo::core::ScaleEstimators::MAD && cfg.InfluenceFuntionType == dvo::core::InfluenceFunctions::Huber);
  assert(std::fabs(cfg.IntensityDerivativeThreshold - 0.01f) < 1e-9f && cfg.IsSane() && cfg.getNumLevels() == 4);
  std::ostringstream os;
  os << cfg;  // dense_tracking.h:215-237
  assert(os.str().find("Scale Estimator = MAD") != std::string::npos && os.str().find("Influence Function = Huber") != std::string::npos);

  // ---- keyframe_graph.cpp: a result with two levels; the coarse one ended by a rejected iteration
  dvo::DenseTracker::Result r;
  for (int l = 0; l < 2; ++l) {
    dvo::DenseTracker::LevelStats ls;
    ls.Id = 3 - l, ls.MaxValidPixels = 4800, ls.ValidPixels = 4000;
    ls.TerminationCriterion = l == 0 ? dvo::DenseTracker::TerminationCriteria::LogLikelihoodDecreased
                                     : dvo::DenseTracker::TerminationCriteria::IncrementTooSmall;
    for (int k = 0; k < 3; ++k) {
      dvo::DenseTracker::IterationStats is;
      is.Id = k, is.ValidConstraints = 3900 - k, is.TDistributionLogLikelihood = -1e4 - k, is.PriorLogLikelihood = 0.0;
      is.EstimateInformation.setZero();
      for (int i = 0; i < 6; ++i) is.EstimateInformation(i, i) = (double)((i + 1) * (k + 1) * (l + 1));
      is.EstimateInformation(0, 5) = is.EstimateInformation(5, 0) = 0.5;  // symmetric, not diagonal
      ls.Iterations.push_back(is);
    }
    r.Statistics.Levels.push_back(ls);
  }
  dvo_slam::printConstraintStatistics(r);
  // non-const accessors (dense_tracking.h:110-111) hand out something assignable
  r.Statistics.Levels.back().LastIteration().ValidConstraints = 7;
  assert(r.Statistics.Levels.back().Iterations.back().ValidConstraints == 7);
  // LogLikelihoodDecreased: the last iteration WITH an increment is the one before the rejected one
  assert(r.Statistics.Levels.front().LastIterationWithIncrement().Id == 1);
  // eigenvalues ascending; for diag(1..6) * 2 with the (0,5) coupling the extreme ones are those of [[2, .5], [.5, 12]]
  dvo::core::Vector6d ev;
  r.Statistics.Levels.front().LastIterationWithIncrement().InformationEigenValues(ev);
  const double lo = 7.0 - std::sqrt(25.0 + 0.25), hi = 7.0 + std::sqrt(25.0 + 0.25);
  assert(std::fabs(ev(0) - lo) < 1e-12 && std::fabs(ev(5) - hi) < 1e-12 && ev(1) == 4.0 && ev(4) == 10.0);
  assert(std::fabs(r.Statistics.Levels.front().LastIterationWithIncrement().InformationConditionNumber() - hi / lo) < 1e-12);

  // ---- surface_pyramid.h (camera_dense_tracking.cpp:235: scale 0.001; benchmark_slam.cpp:77: 1/5000)
  const unsigned short raw[8] = {0, 1, 5000, 65535, 0, 2500, 1000, 7};
  float out[8], out_sse[8];
  dvo::core::SurfacePyramid::convertRawDepthImage(raw, out, 8, 1.0f / 5000.0f);
  dvo::core::SurfacePyramid::convertRawDepthImageSse(raw, out_sse, 8, 1.0f / 5000.0f);
  assert(out[0] != out[0] && out[4] != out[4] && out[2] == 5000.0f * (1.0f / 5000.0f) && out[3] == 65535.0f * (1.0f / 5000.0f));
  for (int i = 0; i < 8; ++i) assert((out[i] != out[i] && out_sse[i] != out_sse[i]) || out[i] == out_sse[i]);
  std::printf("boundary call sites ok\n");
  return 0;
}
