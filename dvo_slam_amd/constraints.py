"""Python mirror of dvo_slam::constraints (loop-closure proposal validation) over the C ABI.

Same names and meaning as the reference classes: Keyframe, ConstraintProposal (createWithIdentity / createWithRelative),
the voters, ConstraintProposalValidator with its fluent Stage builder (constraint_proposal_validator.h,
constraint_proposal_voter.h, constraint_proposal.h), TrackingResultEvaluation (tracking_result_evaluation.h).  validate()
packs everything into plain structs and makes ONE call to dvo_amd_validate_proposals; every stage runs as one GPU batch.
There is no CPU path.
"""
from __future__ import annotations

import time
import ctypes as C

import numpy as np

from . import capi

MAX_VOTERS = 8
ACCEPT, REJECT = 0, 1
EVAL_LOGLIKELIHOOD, EVAL_NORMALIZED_LOGLIKELIHOOD, EVAL_ENTROPY = 0, 1, 2
VOTER_ODOMETRY_CONSTRAINT, VOTER_NAN_RESULT, VOTER_CONSTRAINT_RATIO, VOTER_TRACKING_RESULT_EVALUATION, VOTER_CROSS_VALIDATION = range(5)


class CKeyframe(C.Structure):
    _fields_ = [("id", C.c_int), ("image", C.c_void_p), ("pose", C.c_double * 16), ("evaluation_kind", C.c_int),
                ("evaluation_average", C.c_double), ("evaluation_n", C.c_double)]


class CVoter(C.Structure):
    _fields_ = [("kind", C.c_int), ("threshold", C.c_double)]


class CStage(C.Structure):
    _fields_ = [("id", C.c_int), ("only_keep_best", C.c_int), ("tracking_config", capi.CConfig), ("n_voters", C.c_int),
                ("voters", CVoter * MAX_VOTERS)]


class CVote(C.Structure):
    _fields_ = [("voter_kind", C.c_int), ("reject", C.c_int), ("score", C.c_double), ("value", C.c_double)]


class CProposal(C.Structure):
    _fields_ = [("reference", C.c_int), ("current", C.c_int), ("initial_transformation", C.c_double * 16),
                ("tracking_result", capi.CResult), ("n_votes", C.c_int), ("votes", CVote * MAX_VOTERS),
                ("origin", C.c_int), ("reserved", C.c_int), ("stage_initial_transformation", C.c_double * 16)]


_bound = False


def _lib():
    global _bound
    L = capi.lib()
    if not _bound:
        L.dvo_amd_default_validator_stages.restype = None
        L.dvo_amd_default_validator_stages.argtypes = [C.POINTER(capi.CConfig), C.c_double, C.c_double, C.c_double,
                                                       C.POINTER(CStage)]
        L.dvo_amd_proposals_for_candidates.argtypes = [C.POINTER(CKeyframe), C.c_int, C.c_int, C.POINTER(C.c_int),
                                                       C.POINTER(CProposal)]
        L.dvo_amd_validate_proposals.argtypes = [C.c_void_p, C.c_int, C.POINTER(CKeyframe), C.c_int, C.POINTER(CStage),
                                                 C.c_int, C.POINTER(CProposal), C.POINTER(C.c_int), C.c_int]
        _bound = True
    return L


def _colmajor(T):
    return (C.c_double * 16)(*np.asarray(T, dtype=np.float64).T.reshape(-1))


# ---- tracking_result_evaluation.h ----------------------------------------------------------------------------------
class TrackingResultEvaluation:
    """State of dvo_slam::TrackingResultEvaluation (first_, average_, n_); `kind` picks value(r)."""
    kind = EVAL_LOGLIKELIHOOD

    def __init__(self, first_result: capi.Result):
        self.first = self.value(first_result)
        self.average = self.first
        self.n = 1.0

    def value(self, r: capi.Result) -> float:
        return -r.LogLikelihood

    def add(self, r: capi.Result):
        self.average += self.value(r)
        self.n += 1.0

    def ratioWithFirst(self, r: capi.Result) -> float:
        return self.value(r) / self.first

    def ratioWithAverage(self, r: capi.Result) -> float:
        return self.value(r) / self.average * self.n


class LogLikelihoodTrackingResultEvaluation(TrackingResultEvaluation):
    pass


class NormalizedLogLikelihoodTrackingResultEvaluation(TrackingResultEvaluation):
    kind = EVAL_NORMALIZED_LOGLIKELIHOOD

    def value(self, r):
        with np.errstate(all="ignore"):
            return float(np.float64(-r.LogLikelihood) / np.float64(r.Levels[-1]["Iterations"][-1]["ValidConstraints"]))


class EntropyRatioTrackingResultEvaluation(TrackingResultEvaluation):
    kind = EVAL_ENTROPY

    def value(self, r):
        with np.errstate(all="ignore"):
            return float(np.log(np.float64(np.linalg.det(r.Information))))


class Keyframe:
    """dvo_slam::Keyframe as the validator reads it: id(), image(), pose(), evaluation()."""

    def __init__(self, id: int, image: capi.RgbdImagePyramid, pose, evaluation: TrackingResultEvaluation):
        self.id, self.image, self.pose, self.evaluation = id, image, np.asarray(pose, dtype=np.float64), evaluation


# ---- constraint_proposal.h -------------------------------------------------------------------------------------------
class Vote:
    def __init__(self, c: CVote):
        self.Decision, self.Score, self.Value, self.voter_kind = c.reject, c.score, c.value, c.voter_kind


class ConstraintProposal:
    def __init__(self, reference: Keyframe, current: Keyframe, initial):
        self.Reference, self.Current = reference, current
        self.InitialTransformation = np.asarray(initial, dtype=np.float64)
        self.TrackingResult = None
        self.Votes = []

    @staticmethod
    def createWithIdentity(reference, current):
        return ConstraintProposal(reference, current, np.eye(4))

    @staticmethod
    def createWithRelative(reference, current):
        """InitialTransformation = current.pose^-1 * reference.pose (constraint_proposal.cpp:41-49), evaluated by the library
        so that every binding starts from bit-identical transforms."""
        ckf = (CKeyframe * 2)()
        ckf[0].pose, ckf[1].pose = _colmajor(reference.pose), _colmajor(current.pose)
        out = (CProposal * 2)()
        cand = (C.c_int * 1)(1)
        capi._check(_lib().dvo_amd_proposals_for_candidates(ckf, 0, 1, cand, out), "dvo_amd_proposals_for_candidates")
        return ConstraintProposal(reference, current, np.array(out[1].initial_transformation[:]).reshape(4, 4).T)

    def TotalScore(self) -> float:
        return sum(v.Score for v in self.Votes) if self.Votes else 0.0

    def Accept(self) -> bool:
        return all(v.Decision != REJECT for v in self.Votes)

    def Reject(self) -> bool:
        return any(v.Decision == REJECT for v in self.Votes)


# ---- constraint_proposal_voter.h: plain descriptions, the voting itself happens behind the C ABI ---------------------
class _Voter:
    kind = -1

    def __init__(self, threshold: float = 0.0):
        self.threshold = float(threshold)


class CrossValidationVoter(_Voter):
    kind = VOTER_CROSS_VALIDATION


class TrackingResultEvaluationVoter(_Voter):
    kind = VOTER_TRACKING_RESULT_EVALUATION


class ConstraintRatioVoter(_Voter):
    kind = VOTER_CONSTRAINT_RATIO


class NaNResultVoter(_Voter):
    kind = VOTER_NAN_RESULT


class OdometryConstraintVoter(_Voter):
    kind = VOTER_ODOMETRY_CONSTRAINT


# ---- constraint_proposal_validator.h -----------------------------------------------------------------------------------
class Stage:
    def __init__(self, id: int):
        self.Id, self.OnlyKeepBest, self.TrackingConfig, self.Voters = id, False, capi.Config(), []

    def keepBest(self):
        self.OnlyKeepBest = True
        return self

    def keepAll(self):
        self.OnlyKeepBest = False
        return self

    def trackingConfig(self, cfg: capi.Config):
        self.TrackingConfig = cfg
        return self

    def addVoter(self, v: _Voter):
        if len(self.Voters) >= MAX_VOTERS:
            raise ValueError("too many voters in one stage")
        self.Voters.append(v)
        return self

    def _c(self) -> CStage:
        s = CStage()
        s.id, s.only_keep_best, s.tracking_config, s.n_voters = self.Id, int(self.OnlyKeepBest), self.TrackingConfig._c(), len(self.Voters)
        for i, v in enumerate(self.Voters):
            s.voters[i].kind, s.voters[i].threshold = v.kind, v.threshold
        return s


class ConstraintProposalValidator:
    """validate(proposals) runs all stages in one native call; every stage is one batched GPU alignment."""

    def __init__(self, tracker: capi.DenseTracker | None = None, device: int = 0, max_in_flight: int = 0):
        self.tracker = tracker or capi.DenseTracker(device=device)
        self.max_in_flight = max_in_flight
        self.stages = []

    def createStage(self, id: int) -> Stage:
        self.stages.append(Stage(id))
        return self.stages[-1]

    def validate(self, proposals: list) -> list:
        """In place, like the reference: `proposals` ends up holding the surviving proposals (also returned)."""
        keyframes, index = [], {}
        for p in proposals:
            for kf in (p.Reference, p.Current):
                if id(kf) not in index:
                    index[id(kf)] = len(keyframes)
                    keyframes.append(kf)
        ckf = (CKeyframe * max(len(keyframes), 1))()
        for i, kf in enumerate(keyframes):
            ckf[i].id, ckf[i].image, ckf[i].pose = kf.id, kf.image._h, _colmajor(kf.pose)
            ckf[i].evaluation_kind = kf.evaluation.kind
            ckf[i].evaluation_average, ckf[i].evaluation_n = kf.evaluation.average, kf.evaluation.n
        cst = (CStage * max(len(self.stages), 1))(*[s._c() for s in self.stages])
        cpr = (CProposal * max(len(proposals), 1))()
        for i, p in enumerate(proposals):
            cpr[i].reference, cpr[i].current = index[id(p.Reference)], index[id(p.Current)]
            cpr[i].initial_transformation = _colmajor(p.InitialTransformation)
        n_out = C.c_int()
        t0 = time.perf_counter()
        status = _lib().dvo_amd_validate_proposals(self.tracker._h, len(keyframes), ckf, len(self.stages), cst,
                                                   len(proposals), cpr, C.byref(n_out), self.max_in_flight)
        self.native_ms = (time.perf_counter() - t0) * 1e3  # the library call alone, without this binding's marshalling
        capi._check(status, "dvo_amd_validate_proposals")
        out = []
        for i in range(n_out.value):
            c = cpr[i]
            p = ConstraintProposal(keyframes[c.reference], keyframes[c.current],
                                   np.array(c.initial_transformation[:]).reshape(4, 4).T)
            p.TrackingResult = capi.Result(c.tracking_result, None)
            p.Votes = [Vote(c.votes[k]) for k in range(c.n_votes)]
            p.origin = c.origin  # instrumentation: index of the input proposal it descends from, -(i + 1) for its inverse
            p.stage_initial = np.array(c.stage_initial_transformation[:]).reshape(4, 4).T  # ... what the last stage started from
            out.append(p)
        proposals[:] = out
        return proposals


def createConstraintProposalValidator(frontend_cfg: capi.Config | None = None, min_constraint_ratio: float = 0.2,
                                      ratio_coarse: float = 0.7, ratio_fine: float = 0.9, tracker=None, device: int = 0,
                                      max_in_flight: int = 0) -> ConstraintProposalValidator:
    """KeyframeGraph::createConstraintProposalValidator with configureValidationTracking's tracker configs
    (keyframe_graph.cpp:500-523, 819-838); threshold defaults of dvo_slam/src/config.cpp:38-43."""
    cst = (CStage * 2)()
    c = frontend_cfg._c() if frontend_cfg is not None else None
    _lib().dvo_amd_default_validator_stages(C.byref(c) if c is not None else None, min_constraint_ratio, ratio_coarse,
                                            ratio_fine, cst)
    kinds = {VOTER_CROSS_VALIDATION: CrossValidationVoter, VOTER_TRACKING_RESULT_EVALUATION: TrackingResultEvaluationVoter,
             VOTER_CONSTRAINT_RATIO: ConstraintRatioVoter, VOTER_NAN_RESULT: NaNResultVoter,
             VOTER_ODOMETRY_CONSTRAINT: OdometryConstraintVoter}
    r = ConstraintProposalValidator(tracker, device, max_in_flight)
    for s in cst:
        t = s.tracking_config
        st = r.createStage(s.id).trackingConfig(capi.Config(
            FirstLevel=t.first_level, LastLevel=t.last_level, MaxIterationsPerLevel=t.max_iterations_per_level,
            Precision=t.precision, Mu=t.mu, UseInitialEstimate=bool(t.use_initial_estimate),
            IntensityDerivativeThreshold=t.intensity_derivative_threshold,
            DepthDerivativeThreshold=t.depth_derivative_threshold, SegmentGeometry=t.segment_geometry))
        st.OnlyKeepBest = bool(s.only_keep_best)
        for k in range(s.n_voters):
            st.addVoter(kinds[s.voters[k].kind](s.voters[k].threshold))
    return r


def proposalsForCandidates(keyframe: Keyframe, candidates: list) -> list:
    """The initial proposal list of validateKeyframeConstraintsParallel (keyframe_graph.cpp:577-585)."""
    out = []
    for c in candidates:
        out.append(ConstraintProposal.createWithIdentity(keyframe, c))
        out.append(ConstraintProposal.createWithRelative(keyframe, c))
    return out
