"""CPU restatement of dvo_slam's loop-closure proposal validation -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may import this module.  PARITY UNPINNED: the
reference holds no tests or fixtures for this code and cannot be built here (Eigen / Sophus / Boost / TBB / g2o absent).

Restates, object for object, dvo_slam::constraints::{ConstraintProposal, the five voters, ConstraintProposalValidator}
(constraint_proposal.cpp:31-110, constraint_proposal_voter.cpp:34-186, constraint_proposal_validator.cpp:69-166),
dvo_slam::TrackingResultEvaluation (tracking_result_evaluation.cpp:28-67), the stage set-up of KeyframeGraph
(keyframe_graph.cpp:500-523, 577-585, 819-838) and LevelStats::{Has,Last}IterationWithIncrement
(dense_tracking_config.cpp:138-171), with every match() done by the oracle tracker (oracle.match).
"""
from __future__ import annotations

import numpy as np

from . import oracle as orc

ACCEPT, REJECT = 0, 1
LOGLIKELIHOOD_DECREASED, TOO_FEW_CONSTRAINTS = 2, 3


# ---- tracking_result_evaluation.cpp ---------------------------------------------------------------------------------
class TrackingResultEvaluation:
    def __init__(self, first_result):
        self.first = self.value(first_result)  # :46-51
        self.average = self.first
        self.n = 1.0

    def add(self, r):  # :28-32
        self.average += self.value(r)
        self.n += 1.0

    def ratioWithAverage(self, r):  # :39-42
        return self.value(r) / self.average * self.n


class LogLikelihoodTrackingResultEvaluation(TrackingResultEvaluation):
    kind = 0

    def value(self, r):  # :59-62
        return -r["loglik"]


class NormalizedLogLikelihoodTrackingResultEvaluation(TrackingResultEvaluation):
    kind = 1

    def value(self, r):  # :64-67
        with np.errstate(all="ignore"):  # C++ double division: x/0 is inf or NaN, not an exception
            return float(np.float64(-r["loglik"]) / np.float64(r["levels"][-1]["iterations"][-1]["valid_constraints"]))


class EntropyRatioTrackingResultEvaluation(TrackingResultEvaluation):
    kind = 2

    def value(self, r):  # :54-57
        with np.errstate(all="ignore"):
            return float(np.log(np.float64(np.linalg.det(r["information"]))))


class Keyframe:
    def __init__(self, id, image, pose, evaluation):
        self.id, self.image, self.pose, self.evaluation = id, image, np.asarray(pose, dtype=np.float64), evaluation


# ---- dense_tracking_config.cpp:138-171 ---------------------------------------------------------------------------------
def has_iteration_with_increment(level):
    need = 2 if level["termination"] in (LOGLIKELIHOOD_DECREASED, TOO_FEW_CONSTRAINTS) else 1
    return len(level["iterations"]) >= need


def last_iteration_with_increment(level):
    its = level["iterations"]
    return its[-2] if level["termination"] == LOGLIKELIHOOD_DECREASED else its[-1]


# ---- constraint_proposal.cpp ---------------------------------------------------------------------------------------
class Vote:
    def __init__(self):
        self.Decision, self.Score, self.Value = REJECT, 0.0, 0.0  # constraint_proposal.h:53


class ConstraintProposal:
    def __init__(self, reference, current, initial):
        self.Reference, self.Current = reference, current
        self.InitialTransformation = np.asarray(initial, dtype=np.float64)
        self.TrackingResult = None
        self.Votes = []
        self.origin = None  # test instrumentation: index of the input proposal this one descends from, -(i + 1) for its inverse

    @staticmethod
    def createWithIdentity(reference, current):  # :31-39
        return ConstraintProposal(reference, current, np.eye(4))

    @staticmethod
    def createWithRelative(reference, current):  # :41-49
        return ConstraintProposal(reference, current, np.linalg.inv(current.pose) @ reference.pose)

    def TotalScore(self):  # :55-65
        return sum(v.Score for v in self.Votes) if self.Votes else 0.0

    def Accept(self):  # :67-73
        return all(v.Decision != REJECT for v in self.Votes)

    def Reject(self):  # :75-81
        return any(v.Decision == REJECT for v in self.Votes)

    def createInverseProposal(self):  # :88-96
        inv = ConstraintProposal(self.Current, self.Reference, np.linalg.inv(self.InitialTransformation))
        inv.origin = None if self.origin is None else -(self.origin + 1)
        return inv

    def isConstraintBetweenSameFrames(self, other):  # :98-101
        return ((self.Reference.id == other.Reference.id and self.Current.id == other.Current.id) or
                (self.Reference.id == other.Current.id and self.Current.id == other.Reference.id))


# ---- constraint_proposal_voter.cpp -----------------------------------------------------------------------------------
class Voter:
    def createAdditionalProposals(self, proposals):
        pass

    def removeAdditionalProposals(self, proposals):
        pass


class CrossValidationVoter(Voter):  # :34-99
    def __init__(self, threshold):
        self.TranslationThreshold = threshold
        self.pairs = []

    def createAdditionalProposals(self, proposals):
        for idx in range(len(proposals)):
            other = proposals[idx].createInverseProposal()
            proposals.append(other)
            self.pairs.append((proposals[idx], other))

    def removeAdditionalProposals(self, proposals):
        for first, second in self.pairs:
            worse = second if (first.TotalScore() >= second.TotalScore() and first.Accept()) else first
            for i, p in enumerate(proposals):
                if p is worse:
                    del proposals[i]
                    break
        self.pairs = []

    def vote(self, proposal):
        inverse = None
        for first, second in self.pairs:
            if first is proposal:
                inverse = second
                break
            if second is proposal:
                inverse = first
                break
        diff = inverse.TrackingResult["T"] @ proposal.TrackingResult["T"]
        v = Vote()
        v.Value = float(np.linalg.norm(diff[:3, 3]))
        v.Decision = ACCEPT if v.Value <= self.TranslationThreshold else REJECT
        return v


class TrackingResultEvaluationVoter(Voter):  # :101-119
    def __init__(self, threshold):
        self.RatioThreshold = threshold

    def vote(self, proposal):
        ratio = proposal.Reference.evaluation.ratioWithAverage(proposal.TrackingResult)
        v = Vote()
        v.Decision = ACCEPT if ratio >= self.RatioThreshold else REJECT
        v.Score = v.Value = ratio
        return v


class ConstraintRatioVoter(Voter):  # :121-142
    def __init__(self, threshold):
        self.RatioThreshold = threshold

    def vote(self, proposal):
        l = proposal.TrackingResult["levels"][-1]
        ratio = (last_iteration_with_increment(l)["valid_constraints"] / float(l["valid_pixels"])
                 if has_iteration_with_increment(l) else 0.0)
        v = Vote()
        v.Value = ratio
        v.Decision = ACCEPT if ratio >= self.RatioThreshold else REJECT
        return v


class NaNResultVoter(Voter):  # :144-162
    def vote(self, proposal):
        v = Vote()
        v.Value = float(proposal.TrackingResult["is_nan"])
        v.Decision = REJECT if proposal.TrackingResult["is_nan"] else ACCEPT
        return v


class OdometryConstraintVoter(Voter):  # :164-184
    def vote(self, proposal):
        is_odometry = abs(proposal.Reference.id - proposal.Current.id) <= 1
        v = Vote()
        v.Value = float(is_odometry)
        v.Decision = REJECT if is_odometry else ACCEPT
        return v


# ---- constraint_proposal_validator.cpp -------------------------------------------------------------------------------
class Stage:
    def __init__(self, id):
        self.Id, self.OnlyKeepBest, self.TrackingConfig, self.Voters = id, False, None, []

    def keepBest(self):
        self.OnlyKeepBest = True
        return self

    def keepAll(self):
        self.OnlyKeepBest = False
        return self

    def trackingConfig(self, cfg):
        self.TrackingConfig = cfg
        return self

    def addVoter(self, v):
        self.Voters.append(v)
        return self


class ConstraintProposalValidator:
    def __init__(self):
        self.stages = []
        self.n_matches = 0
        self.history = []  # test instrumentation: (stage id, origin, reference id, current id, initial transformation, result)
                           # of EVERY alignment, also of the proposals that do not survive

    def createStage(self, id):
        self.stages.append(Stage(id))
        return self.stages[-1]

    def validate(self, proposals):  # :69-102, in place
        for i, p in enumerate(proposals):
            p.origin = i
        for stage in self.stages:
            for p in proposals:
                p.Votes = []
                p.TrackingResult = None
            self._validate_stage(stage, proposals)
            proposals[:] = [p for p in proposals if not p.Reject()]
            if stage.OnlyKeepBest:
                self.keepBest(proposals)
            for p in proposals:
                p.InitialTransformation = np.linalg.inv(p.TrackingResult["T"])
        return proposals

    @staticmethod
    def keepBest(proposals):  # :104-130
        i = 0
        while i < len(proposals):
            k = i + 1
            while k < len(proposals):
                if proposals[i].isConstraintBetweenSameFrames(proposals[k]):
                    if proposals[k].TotalScore() > proposals[i].TotalScore():
                        proposals[i], proposals[k] = proposals[k], proposals[i]
                    del proposals[k]
                else:
                    k += 1
            i += 1

    def _validate_stage(self, stage, proposals):  # :132-163
        for v in stage.Voters:
            v.createAdditionalProposals(proposals)
        for p in proposals:
            p.TrackingResult = orc.match(stage.TrackingConfig, p.Reference.image, p.Current.image, p.InitialTransformation)
            self.n_matches += 1
            self.history.append((stage.Id, p.origin, p.Reference.id, p.Current.id, p.InitialTransformation.copy(), p.TrackingResult))
        for p in proposals:
            for v in stage.Voters:
                p.Votes.append(v.vote(p))
                if p.Votes[-1].Decision == REJECT:
                    break
        for v in reversed(stage.Voters):
            v.removeAdditionalProposals(proposals)


def create_constraint_proposal_validator(frontend_cfg=None, min_constraint_ratio=0.2, ratio_coarse=0.7, ratio_fine=0.9,
                                         rcp_mode=orc.RCP_EXACT):
    """KeyframeGraph::createConstraintProposalValidator + configureValidationTracking (keyframe_graph.cpp:500-523, 819-838);
    threshold defaults of dvo_slam/src/config.cpp:38-43."""
    def tracker_cfg(last_level):
        kw = dict(first_level=3, last_level=last_level, use_initial_estimate=1, rcp_mode=rcp_mode)
        if frontend_cfg is not None:
            kw.update(precision=frontend_cfg.precision, mu=frontend_cfg.mu,
                      intensity_derivative_threshold=frontend_cfg.intensity_derivative_threshold,
                      depth_derivative_threshold=frontend_cfg.depth_derivative_threshold)
        return orc.default_config(**kw)

    r = ConstraintProposalValidator()
    (r.createStage(1).trackingConfig(tracker_cfg(3)).keepAll()
        .addVoter(OdometryConstraintVoter()).addVoter(NaNResultVoter())
        .addVoter(ConstraintRatioVoter(min_constraint_ratio)).addVoter(TrackingResultEvaluationVoter(ratio_coarse))
        .addVoter(CrossValidationVoter(1.0)))
    (r.createStage(2).trackingConfig(tracker_cfg(1)).keepBest()
        .addVoter(NaNResultVoter()).addVoter(ConstraintRatioVoter(min_constraint_ratio))
        .addVoter(TrackingResultEvaluationVoter(ratio_fine)))
    return r


def proposals_for_candidates(keyframe, candidates):
    """validateKeyframeConstraintsParallel's initial list, keyframe_graph.cpp:577-585."""
    out = []
    for c in candidates:
        out.append(ConstraintProposal.createWithIdentity(keyframe, c))
        out.append(ConstraintProposal.createWithRelative(keyframe, c))
    return out
