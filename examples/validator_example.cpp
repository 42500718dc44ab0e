// Loop-closure validation written the way dvo_slam/src/keyframe_graph.cpp:500-593 uses the reference API: keyframes, two
// proposals per candidate, the two-stage validator, survivors printed.  Input: a directory written by
// tests/test_cpp_adaptor.py holding frames.txt (one line per keyframe: id, 16 pose values row-major) and the raw float32
// planes <id>_i.f32 / <id>_z.f32; the first keyframe is matched against all others.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "dvo_amd/constraints.hpp"

using namespace dvo_slam;
using namespace dvo_slam::constraints;

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

int main(int argc, char **argv) {
  if (argc < 11) {
    std::fprintf(stderr, "usage: %s dir w h fx fy ox oy min_constraint_ratio ratio_coarse ratio_fine\n", argv[0]);
    return 2;
  }
  const std::string dir = argv[1];
  const int w = std::atoi(argv[2]), h = std::atoi(argv[3]);
  dvo::core::IntrinsicMatrix K = dvo::core::IntrinsicMatrix::create((float)std::atof(argv[4]), (float)std::atof(argv[5]),
                                                                      (float)std::atof(argv[6]), (float)std::atof(argv[7]));
  const double min_ratio = std::atof(argv[8]), coarse = std::atof(argv[9]), fine = std::atof(argv[10]);
  dvo::core::RgbdCameraPyramid camera(w, h, K);

  // the odometry tracker that seeds every keyframe's evaluation (keyframe_tracker.cpp:88-96): a frame against itself here
  dvo::DenseTracker::Config odo = dvo::DenseTracker::getDefaultConfig();
  dvo::DenseTracker odometry(odo);

  KeyframeVector keyframes;
  std::ifstream list((dir + "/frames.txt").c_str());
  std::string line;
  while (std::getline(list, line)) {
    std::istringstream in(line);
    int id;
    in >> id;
    dvo::core::AffineTransformd pose;
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) in >> dvo::core::data(pose)[c * 4 + r];
    std::ostringstream base;
    base << dir << "/" << id;
    std::vector<float> I = read_plane(base.str() + "_i.f32", (size_t)w * h), Z = read_plane(base.str() + "_z.f32", (size_t)w * h);
    dvo::core::RgbdImagePyramidPtr image = camera.create(I.data(), Z.data());
    dvo::DenseTracker::Result first;
    odometry.match(*image, *image, first);
    KeyframePtr kf(new Keyframe());
    kf->id(id).image(image).pose(pose).evaluation(
        TrackingResultEvaluation::ConstPtr(new LogLikelihoodTrackingResultEvaluation(first)));
    keyframes.push_back(kf);
  }
  if (keyframes.size() < 2) return 2;

  // configureValidationTracking + createConstraintProposalValidator, keyframe_graph.cpp:500-523,819-838
  dvo::DenseTracker::Config validation = dvo::DenseTracker::getDefaultConfig(), constraint = validation;
  validation.FirstLevel = 3, validation.LastLevel = 3, validation.UseInitialEstimate = true;
  constraint.FirstLevel = 3, constraint.LastLevel = 1, constraint.UseInitialEstimate = true;
  ConstraintProposalValidator validator;
  validator.createStage(1)
      .trackingConfig(validation)
      .keepAll()
      .addVoter(new OdometryConstraintVoter())
      .addVoter(new NaNResultVoter())
      .addVoter(new ConstraintRatioVoter(min_ratio))
      .addVoter(new TrackingResultEvaluationVoter(coarse))
      .addVoter(new CrossValidationVoter(1.0));
  validator.createStage(2)
      .trackingConfig(constraint)
      .keepBest()
      .addVoter(new NaNResultVoter())
      .addVoter(new ConstraintRatioVoter(min_ratio))
      .addVoter(new TrackingResultEvaluationVoter(fine));

  // validateKeyframeConstraintsParallel, keyframe_graph.cpp:577-585
  ConstraintProposalVector proposals;
  for (size_t i = 1; i < keyframes.size(); ++i) {
    proposals.push_back(ConstraintProposal::createWithIdentity(keyframes[0], keyframes[i]));
    proposals.push_back(ConstraintProposal::createWithRelative(keyframes[0], keyframes[i]));
  }
  validator.validate(proposals);

  std::printf("survivors %zu\n", proposals.size());
  for (size_t i = 0; i < proposals.size(); ++i) {
    const ConstraintProposal &p = *proposals[i];
    const double *T = dvo::core::data(p.TrackingResult.Transformation);
    std::printf("proposal %d %d score %.9g votes %zu accept %d", p.Reference->id(), p.Current->id(), p.TotalScore(), p.Votes.size(),
                p.Accept() ? 1 : 0);
    for (int k = 0; k < 16; ++k) std::printf(" %.17g", T[(k % 4) * 4 + k / 4]);  // row-major
    std::printf("\n");
  }
  return 0;
}
