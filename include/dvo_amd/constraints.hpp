// constraints.hpp -- header-only C++ adaptor over the C ABI for the callers either side of the tracking path.
//
// Re-declares, name for name, what dvo_slam's back end uses to validate loop-closure proposals:
//   dvo_slam::TrackingResultEvaluation (+ LogLikelihood / NormalizedLogLikelihood / EntropyRatio)
//                                                        dvo_slam/include/dvo_slam/tracking_result_evaluation.h:31-77
//   dvo_slam::Keyframe {id, image, pose, evaluation}     dvo_slam/include/dvo_slam/keyframe.h
//   dvo_slam::constraints::ConstraintProposal            .../constraints/constraint_proposal.h:34-90
//   dvo_slam::constraints::{CrossValidation, TrackingResultEvaluation, ConstraintRatio, NaNResult, OdometryConstraint}Voter
//                                                        .../constraints/constraint_proposal_voter.h
//   dvo_slam::constraints::ConstraintProposalValidator   .../constraints/constraint_proposal_validator.h
// and the per-frame step of dvo_slam::LocalTracker::update (local_tracker.cpp:170-191) as trackFrame().
// validate() makes ONE call into libdvo_amd.so (dvo_amd_validate_proposals): every stage is one batched GPU alignment.
// The voters are plain descriptions here; the voting itself happens behind the C ABI.
#ifndef DVO_AMD_CONSTRAINTS_HPP_
#define DVO_AMD_CONSTRAINTS_HPP_

#include <map>
#include <memory>
#include <vector>

#include "dense_tracking.hpp"

namespace dvo_slam {

// tracking_result_evaluation.h:31-77
class TrackingResultEvaluation {
 public:
  typedef std::shared_ptr<TrackingResultEvaluation> Ptr;
  typedef std::shared_ptr<const TrackingResultEvaluation> ConstPtr;
  virtual ~TrackingResultEvaluation() {}
  virtual void add(const dvo::DenseTracker::Result &r) {
    average_ += value(r);
    n_ += 1.0;
  }
  virtual double ratioWithFirst(const dvo::DenseTracker::Result &r) const { return value(r) / first_; }
  virtual double ratioWithAverage(const dvo::DenseTracker::Result &r) const { return value(r) / average_ * n_; }
  virtual double value(const dvo::DenseTracker::Result &r) const = 0;
  virtual int kind() const = 0;  // dvo_amd_evaluation_kind
  double average() const { return average_; }
  double n() const { return n_; }

 protected:
  TrackingResultEvaluation() : first_(0.0), average_(0.0), n_(1.0) {}
  void init(double first) { first_ = first, average_ = first, n_ = 1.0; }
  double first_, average_, n_;
};

class LogLikelihoodTrackingResultEvaluation : public TrackingResultEvaluation {
 public:
  explicit LogLikelihoodTrackingResultEvaluation(const dvo::DenseTracker::Result &r) { init(value(r)); }
  double value(const dvo::DenseTracker::Result &r) const override { return -r.LogLikelihood; }
  int kind() const override { return DVO_AMD_EVAL_LOGLIKELIHOOD; }
};

class NormalizedLogLikelihoodTrackingResultEvaluation : public TrackingResultEvaluation {
 public:
  explicit NormalizedLogLikelihoodTrackingResultEvaluation(const dvo::DenseTracker::Result &r) { init(value(r)); }
  double value(const dvo::DenseTracker::Result &r) const override {
    return -r.LogLikelihood / (double)r.Statistics.Levels.back().Iterations.back().ValidConstraints;
  }
  int kind() const override { return DVO_AMD_EVAL_NORMALIZED_LOGLIKELIHOOD; }
};

class EntropyRatioTrackingResultEvaluation : public TrackingResultEvaluation {
 public:
  explicit EntropyRatioTrackingResultEvaluation(const dvo::DenseTracker::Result &r) { init(value(r)); }
  double value(const dvo::DenseTracker::Result &r) const override {  // log(det(Information)), LU with partial pivoting
    double m[6][6];
    for (int row = 0; row < 6; ++row)
      for (int col = 0; col < 6; ++col) m[row][col] = r.Information(row, col);
    double det = 1.0;
    for (int k = 0; k < 6; ++k) {
      int piv = k;
      for (int row = k + 1; row < 6; ++row)
        if (std::fabs(m[row][k]) > std::fabs(m[piv][k])) piv = row;
      if (m[piv][k] == 0.0) return std::log(0.0);
      if (piv != k) {
        for (int col = 0; col < 6; ++col) std::swap(m[piv][col], m[k][col]);
        det = -det;
      }
      det *= m[k][k];
      for (int row = k + 1; row < 6; ++row) {
        const double f = m[row][k] / m[k][k];
        for (int col = k; col < 6; ++col) m[row][col] -= f * m[k][col];
      }
    }
    return std::log(det);
  }
  int kind() const override { return DVO_AMD_EVAL_ENTROPY; }
};

// keyframe.h: the accessors the validator reads
class Keyframe {
 public:
  Keyframe() : id_(-1) {}
  int id() const { return id_; }
  Keyframe &id(int v) { id_ = v; return *this; }
  dvo::core::RgbdImagePyramid::Ptr image() const { return image_; }
  Keyframe &image(const dvo::core::RgbdImagePyramid::Ptr &v) { image_ = v; return *this; }
  const dvo::core::AffineTransformd &pose() const { return pose_; }
  Keyframe &pose(const dvo::core::AffineTransformd &v) { pose_ = v; return *this; }
  TrackingResultEvaluation::ConstPtr evaluation() const { return evaluation_; }
  Keyframe &evaluation(const TrackingResultEvaluation::ConstPtr &v) { evaluation_ = v; return *this; }

 private:
  int id_;
  dvo::core::RgbdImagePyramid::Ptr image_;
  dvo::core::AffineTransformd pose_;
  TrackingResultEvaluation::ConstPtr evaluation_;
};
typedef std::shared_ptr<Keyframe> KeyframePtr;
typedef std::vector<KeyframePtr> KeyframeVector;

namespace constraints {

// constraint_proposal.h:34-90
struct ConstraintProposal {
  struct Vote {
    enum Enum { Accept, Reject };
    Enum Decision;
    double Score;
    double Value;  // the quantity the voter tested (the reference prints it into Vote::Reason)
    int VoterKind;
    Vote() : Decision(Reject), Score(0.0), Value(0.0), VoterKind(-1) {}
  };
  typedef std::vector<Vote> VoteVector;

  KeyframePtr Reference, Current;
  dvo::core::AffineTransformd InitialTransformation;
  dvo::DenseTracker::Result TrackingResult;
  VoteVector Votes;

  static std::shared_ptr<ConstraintProposal> createWithIdentity(const KeyframePtr &reference, const KeyframePtr &current) {
    std::shared_ptr<ConstraintProposal> p(new ConstraintProposal());
    p->Reference = reference, p->Current = current;
    p->InitialTransformation.setIdentity();
    return p;
  }
  // InitialTransformation = current->pose().inverse() * reference->pose()  (constraint_proposal.cpp:41-49); evaluated by the
  // library so that the adaptor needs no matrix algebra of its own
  static std::shared_ptr<ConstraintProposal> createWithRelative(const KeyframePtr &reference, const KeyframePtr &current) {
    std::shared_ptr<ConstraintProposal> p(new ConstraintProposal());
    p->Reference = reference, p->Current = current;
    dvo_amd_keyframe kf[2];
    std::memset(kf, 0, sizeof(kf));
    std::memcpy(kf[0].pose, dvo::core::data(reference->pose()), sizeof(kf[0].pose));
    std::memcpy(kf[1].pose, dvo::core::data(current->pose()), sizeof(kf[1].pose));
    const int candidate = 1;
    dvo_amd_constraint_proposal two[2];
    dvo::detail::check(dvo_amd_proposals_for_candidates(kf, 0, 1, &candidate, two), "ConstraintProposal::createWithRelative");
    std::memcpy(dvo::core::data(p->InitialTransformation), two[1].initial_transformation, sizeof(two[1].initial_transformation));
    return p;
  }
  double TotalScore() const {
    double s = 0.0;
    for (VoteVector::const_iterator it = Votes.begin(); it != Votes.end(); ++it) s += it->Score;
    return s;
  }
  bool Accept() const {
    for (VoteVector::const_iterator it = Votes.begin(); it != Votes.end(); ++it)
      if (it->Decision == Vote::Reject) return false;
    return true;
  }
  bool Reject() const { return !Accept(); }
  void clearVotes() { Votes.clear(); }
};
typedef std::shared_ptr<ConstraintProposal> ConstraintProposalPtr;
typedef std::vector<ConstraintProposalPtr> ConstraintProposalVector;

// constraint_proposal_voter.h
struct ConstraintProposalVoter {
  virtual ~ConstraintProposalVoter() {}
  virtual int kind() const = 0;  // dvo_amd_voter_kind
  virtual double threshold() const { return 0.0; }
};
typedef std::shared_ptr<ConstraintProposalVoter> ConstraintProposalVoterPtr;
struct CrossValidationVoter : ConstraintProposalVoter {
  double TranslationThreshold;
  explicit CrossValidationVoter(double t) : TranslationThreshold(t) {}
  int kind() const override { return DVO_AMD_VOTER_CROSS_VALIDATION; }
  double threshold() const override { return TranslationThreshold; }
};
struct TrackingResultEvaluationVoter : ConstraintProposalVoter {
  double RatioThreshold;
  explicit TrackingResultEvaluationVoter(double t) : RatioThreshold(t) {}
  int kind() const override { return DVO_AMD_VOTER_TRACKING_RESULT_EVALUATION; }
  double threshold() const override { return RatioThreshold; }
};
struct ConstraintRatioVoter : ConstraintProposalVoter {
  double RatioThreshold;
  explicit ConstraintRatioVoter(double t) : RatioThreshold(t) {}
  int kind() const override { return DVO_AMD_VOTER_CONSTRAINT_RATIO; }
  double threshold() const override { return RatioThreshold; }
};
struct NaNResultVoter : ConstraintProposalVoter {
  int kind() const override { return DVO_AMD_VOTER_NAN_RESULT; }
};
struct OdometryConstraintVoter : ConstraintProposalVoter {
  int kind() const override { return DVO_AMD_VOTER_ODOMETRY_CONSTRAINT; }
};

// constraint_proposal_validator.h
class ConstraintProposalValidator {
 public:
  struct Stage {
    int Id;
    bool OnlyKeepBest;
    dvo::DenseTracker::Config TrackingConfig;
    std::vector<ConstraintProposalVoterPtr> Voters;
    explicit Stage(int id) : Id(id), OnlyKeepBest(false) {}
    Stage &keepBest() { OnlyKeepBest = true; return *this; }
    Stage &keepAll() { OnlyKeepBest = false; return *this; }
    Stage &trackingConfig(const dvo::DenseTracker::Config &cfg) { TrackingConfig = cfg; return *this; }
    Stage &addVoter(ConstraintProposalVoter *v) { Voters.push_back(ConstraintProposalVoterPtr(v)); return *this; }
  };

  explicit ConstraintProposalValidator(int device = 0, int max_in_flight = 0) : tracker_(dvo::DenseTracker::getDefaultConfig(), device), max_in_flight_(max_in_flight) {}

  Stage &createStage(int id) {
    stages_.push_back(Stage(id));
    return stages_.back();
  }

  // in place, like the reference: on return `proposals` holds the surviving proposals in the reference's order
  void validate(ConstraintProposalVector &proposals, bool /*debug*/ = false) {
    std::vector<KeyframePtr> keyframes;
    std::map<const Keyframe *, int> index;
    for (size_t i = 0; i < proposals.size(); ++i) {
      const KeyframePtr both[2] = {proposals[i]->Reference, proposals[i]->Current};
      for (int k = 0; k < 2; ++k)
        if (index.find(both[k].get()) == index.end()) {
          index[both[k].get()] = (int)keyframes.size();
          keyframes.push_back(both[k]);
        }
    }
    size_t need_levels = 1;
    for (size_t s = 0; s < stages_.size(); ++s) need_levels = std::max(need_levels, stages_[s].TrackingConfig.getNumLevels());
    std::vector<dvo_amd_keyframe> ckf(std::max<size_t>(keyframes.size(), 1));
    for (size_t i = 0; i < keyframes.size(); ++i) {
      const Keyframe &kf = *keyframes[i];
      kf.image()->build(need_levels);
      std::memset(&ckf[i], 0, sizeof(ckf[i]));
      ckf[i].id = kf.id(), ckf[i].image = kf.image()->handle();
      std::memcpy(ckf[i].pose, dvo::core::data(kf.pose()), sizeof(ckf[i].pose));
      ckf[i].evaluation_kind = kf.evaluation()->kind();
      ckf[i].evaluation_average = kf.evaluation()->average(), ckf[i].evaluation_n = kf.evaluation()->n();
    }
    std::vector<dvo_amd_validator_stage> cst(std::max<size_t>(stages_.size(), 1));
    for (size_t s = 0; s < stages_.size(); ++s) {
      std::memset(&cst[s], 0, sizeof(cst[s]));
      cst[s].id = stages_[s].Id, cst[s].only_keep_best = stages_[s].OnlyKeepBest ? 1 : 0;
      cst[s].tracking_config = to_c(stages_[s].TrackingConfig);
      if (stages_[s].Voters.size() > DVO_AMD_MAX_VOTERS) throw dvo::DvoAmdError(DVO_AMD_ERR_INVALID_ARGUMENT, "Stage::addVoter");
      cst[s].n_voters = (int)stages_[s].Voters.size();
      for (size_t v = 0; v < stages_[s].Voters.size(); ++v)
        cst[s].voters[v].kind = stages_[s].Voters[v]->kind(), cst[s].voters[v].threshold = stages_[s].Voters[v]->threshold();
    }
    std::vector<dvo_amd_constraint_proposal> cpr(std::max<size_t>(proposals.size(), 1));
    for (size_t i = 0; i < proposals.size(); ++i) {
      std::memset(&cpr[i], 0, sizeof(cpr[i]));
      cpr[i].reference = index[proposals[i]->Reference.get()], cpr[i].current = index[proposals[i]->Current.get()];
      std::memcpy(cpr[i].initial_transformation, dvo::core::data(proposals[i]->InitialTransformation), sizeof(double) * 16);
    }
    int n_out = 0;
    dvo::detail::check(dvo_amd_validate_proposals(tracker_.handle(), (int)keyframes.size(), ckf.data(), (int)stages_.size(),
                                                  cst.data(), (int)proposals.size(), cpr.data(), &n_out, max_in_flight_),
                       "ConstraintProposalValidator::validate");
    ConstraintProposalVector out;
    for (int i = 0; i < n_out; ++i) {
      const dvo_amd_constraint_proposal &c = cpr[(size_t)i];
      ConstraintProposalPtr p(new ConstraintProposal());
      p->Reference = keyframes[(size_t)c.reference], p->Current = keyframes[(size_t)c.current];
      std::memcpy(dvo::core::data(p->InitialTransformation), c.initial_transformation, sizeof(double) * 16);
      std::memcpy(dvo::core::data(p->TrackingResult.Transformation), c.tracking_result.transformation, sizeof(double) * 16);
      std::memcpy(dvo::core::data(p->TrackingResult.Information), c.tracking_result.information, sizeof(double) * 36);
      p->TrackingResult.LogLikelihood = c.tracking_result.loglik;
      for (int l = 0; l < c.tracking_result.n_levels; ++l) {  // per-iteration statistics are not carried through the batch
        dvo::DenseTracker::LevelStats ls;
        ls.Id = (size_t)c.tracking_result.levels[l].id;
        ls.MaxValidPixels = (size_t)c.tracking_result.levels[l].max_valid_pixels;
        ls.ValidPixels = (size_t)c.tracking_result.levels[l].valid_pixels;
        ls.TerminationCriterion = (dvo::DenseTracker::TerminationCriteria::Enum)c.tracking_result.levels[l].termination;
        p->TrackingResult.Statistics.Levels.push_back(ls);
      }
      for (int v = 0; v < c.n_votes; ++v) {
        ConstraintProposal::Vote vote;
        vote.Decision = c.votes[v].reject ? ConstraintProposal::Vote::Reject : ConstraintProposal::Vote::Accept;
        vote.Score = c.votes[v].score, vote.Value = c.votes[v].value, vote.VoterKind = c.votes[v].voter_kind;
        p->Votes.push_back(vote);
      }
      out.push_back(p);
    }
    proposals.swap(out);
  }

 private:
  static dvo_amd_config to_c(const dvo::DenseTracker::Config &c) {
    dvo_amd_config o;
    o.first_level = c.FirstLevel, o.last_level = c.LastLevel, o.max_iterations_per_level = c.MaxIterationsPerLevel;
    o.precision = c.Precision, o.mu = c.Mu, o.use_initial_estimate = c.UseInitialEstimate ? 1 : 0;
    o.intensity_derivative_threshold = c.IntensityDerivativeThreshold;
    o.depth_derivative_threshold = c.DepthDerivativeThreshold;
    o.segment_geometry = c.SegmentGeometry, o.reserved = 0;
    return o;
  }
  dvo::DenseTracker tracker_;
  int max_in_flight_;
  std::vector<Stage> stages_;
};
typedef std::shared_ptr<ConstraintProposalValidator> ConstraintProposalValidatorPtr;

}  // namespace constraints

// The two alignments of LocalTracker::update (local_tracker.cpp:170-186) as one two-pair batch, with the inputs of the
// accept callbacks of KeyframeTracker (keyframe_tracker.cpp:105-190) in `criteria`.
inline void trackFrame(dvo::DenseTracker &tracker, dvo::core::RgbdImagePyramid &keyframe, dvo::core::RgbdImagePyramid &last_frame,
                       dvo::core::RgbdImagePyramid &frame, const dvo::core::AffineTransformd &last_keyframe_pose,
                       dvo::DenseTracker::Result &r_keyframe, dvo::DenseTracker::Result &r_odometry,
                       dvo_amd_frame_criteria &criteria) {
  const size_t levels = tracker.configuration().getNumLevels();
  keyframe.build(levels), last_frame.build(levels), frame.build(levels);
  dvo_amd_result rk, ro;
  std::memset(&rk, 0, sizeof(rk));
  std::memset(&ro, 0, sizeof(ro));
  dvo::detail::check(dvo_amd_track_frame(tracker.handle(), keyframe.handle(), last_frame.handle(), frame.handle(),
                                         dvo::core::data(last_keyframe_pose), &rk, &ro, &criteria),
                     "trackFrame");
  dvo::DenseTracker::Result *out[2] = {&r_keyframe, &r_odometry};
  const dvo_amd_result *in[2] = {&rk, &ro};
  for (int k = 0; k < 2; ++k) {
    std::memcpy(dvo::core::data(out[k]->Transformation), in[k]->transformation, sizeof(double) * 16);
    std::memcpy(dvo::core::data(out[k]->Information), in[k]->information, sizeof(double) * 36);
    out[k]->LogLikelihood = in[k]->loglik;
  }
}

}  // namespace dvo_slam

#endif  // DVO_AMD_CONSTRAINTS_HPP_
