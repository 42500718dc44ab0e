"""Loop-closure proposal validation.

CPU part: the oracle restatement (oracle/validator.py) against the reference's documented behaviour, and the host-only
helpers of the C ABI against the oracle.  GPU part (-m gpu): dvo_amd_validate_proposals against the oracle on the same
scenario, for several voter thresholds and evaluation kinds.  PARITY UNPINNED (the reference holds no fixtures for this).
"""
import ctypes as C

import numpy as np
import pytest

import validator_scenario as S

W, H = 320, 240
PERMISSIVE = dict(min_constraint_ratio=0.0, ratio_coarse=-1e300, ratio_fine=-1e300)


@pytest.fixture(scope="module")
def V(orc):
    from oracle import validator

    return validator


@pytest.fixture(scope="module")
def scenario(orc, V, synth):
    return S.oracle_keyframes(orc, V, synth, W, H, 5)


def _rel_true(p):
    """expected TrackingResult.Transformation of a proposal (cur <- ref, dense_tracking.cpp:371)"""
    return p.Current.pose_true @ np.linalg.inv(p.Reference.pose_true)


# ---------------------------------------------------------------------------------------------------------------------
# oracle behaviour
# ---------------------------------------------------------------------------------------------------------------------
def test_stage_setup_follows_keyframe_graph(V, orc):
    v = V.create_constraint_proposal_validator(min_constraint_ratio=0.2, ratio_coarse=0.7, ratio_fine=0.9)
    s1, s2 = v.stages
    assert (s1.Id, s1.OnlyKeepBest, s2.Id, s2.OnlyKeepBest) == (1, False, 2, True)
    assert [type(x).__name__ for x in s1.Voters] == ["OdometryConstraintVoter", "NaNResultVoter", "ConstraintRatioVoter",
                                                     "TrackingResultEvaluationVoter", "CrossValidationVoter"]
    assert [type(x).__name__ for x in s2.Voters] == ["NaNResultVoter", "ConstraintRatioVoter", "TrackingResultEvaluationVoter"]
    assert s1.Voters[4].TranslationThreshold == 1.0 and s1.Voters[3].RatioThreshold == 0.7 and s2.Voters[2].RatioThreshold == 0.9
    assert (s1.TrackingConfig.first_level, s1.TrackingConfig.last_level) == (3, 3)
    assert (s2.TrackingConfig.first_level, s2.TrackingConfig.last_level) == (3, 1)
    assert s1.TrackingConfig.use_initial_estimate == 1 and s2.TrackingConfig.use_initial_estimate == 1


def test_oracle_validation_of_the_scenario(V, synth, scenario):
    key, cands = scenario
    val = V.create_constraint_proposal_validator(**PERMISSIVE)
    props = V.proposals_for_candidates(key, cands)
    assert len(props) == 2 * len(cands)
    assert np.array_equal(props[0].InitialTransformation, np.eye(4))
    assert np.allclose(props[1].InitialTransformation, np.linalg.inv(cands[0].pose) @ key.pose)
    out = val.validate(props)
    assert out is props
    # stage 1 aligned every proposal and its inverse, stage 2 the survivors
    n_ok = len(cands) - 2  # the id-neighbour and the depth-less frame are thrown out in stage 1
    assert val.n_matches == 4 * len(cands) + 2 * n_ok
    pairs = [frozenset((p.Reference.id, p.Current.id)) for p in out]
    assert len(pairs) == len(set(pairs)) == n_ok  # keepBest: one constraint per pair of frames
    assert frozenset((100, 101)) not in pairs and frozenset((100, 70)) not in pairs
    for p in out:
        assert p.Accept() and not p.Reject() and len(p.Votes) == 3
        assert p.TotalScore() == p.Votes[2].Score
        assert np.allclose(p.InitialTransformation, np.linalg.inv(p.TrackingResult["T"]))
        if 60 not in (p.Reference.id, p.Current.id):  # same scene: the estimate is the true relative pose
            assert synth.pose_error(p.TrackingResult["T"], _rel_true(p)) < 5e-3


def test_decoys_are_rejected_by_the_expected_voter(V, scenario):
    key, cands = scenario
    stage = V.create_constraint_proposal_validator(**PERMISSIVE).stages[0]
    neighbour = next(c for c in cands if c.id == 101)
    no_depth = next(c for c in cands if c.id == 70)
    props = [V.ConstraintProposal.createWithIdentity(key, neighbour), V.ConstraintProposal.createWithIdentity(key, no_depth)]
    val = V.ConstraintProposalValidator()
    val._validate_stage(stage, props)
    assert len(props) == 2  # one of every proposal / inverse pair is removed again
    assert len(props[0].Votes) == 1 and props[0].Votes[0].Decision == V.REJECT  # early abort after the odometry voter
    assert len(props[1].Votes) == 2 and props[1].Votes[1].Decision == V.REJECT and props[1].TrackingResult["is_nan"]
    # the inverse replaces a rejected original (CrossValidationVoter::removeAdditionalProposals keeps `second` then)
    assert props[0].Reference.id == 101 and props[1].Reference.id == 70


def test_strict_thresholds_reject_everything(V, scenario):
    key, cands = scenario
    val = V.create_constraint_proposal_validator(min_constraint_ratio=0.2, ratio_coarse=1e9, ratio_fine=0.9)
    assert val.validate(V.proposals_for_candidates(key, cands)) == []


def test_keep_best_keeps_the_highest_score_at_the_first_position(V):
    kf = [V.Keyframe(i, None, np.eye(4), None) for i in range(4)]

    def prop(a, b, score):
        p = V.ConstraintProposal(kf[a], kf[b], np.eye(4))
        v = V.Vote()
        v.Decision, v.Score = V.ACCEPT, score
        p.Votes = [v]
        return p

    ps = [prop(0, 1, 0.3), prop(0, 2, 0.9), prop(1, 0, 0.8), prop(0, 1, 0.5), prop(2, 0, 0.9), prop(3, 0, 0.1)]
    V.ConstraintProposalValidator.keepBest(ps)
    assert [(p.Reference.id, p.Current.id, p.TotalScore()) for p in ps] == [(1, 0, 0.8), (0, 2, 0.9), (3, 0, 0.1)]


def test_level_stats_helpers(V):
    its = [dict(valid_constraints=10), dict(valid_constraints=20), dict(valid_constraints=30)]
    assert V.last_iteration_with_increment(dict(termination=2, iterations=its))["valid_constraints"] == 20
    assert V.last_iteration_with_increment(dict(termination=1, iterations=its))["valid_constraints"] == 30
    assert not V.has_iteration_with_increment(dict(termination=3, iterations=its[:1]))
    assert V.has_iteration_with_increment(dict(termination=0, iterations=its[:1]))


# ---------------------------------------------------------------------------------------------------------------------
# host-only helpers of the C ABI (no GPU needed)
# ---------------------------------------------------------------------------------------------------------------------
def test_default_stages_of_the_c_abi_equal_the_oracle_setup(V):
    from dvo_slam_amd import capi, constraints as Cn

    cst = (Cn.CStage * 2)()
    front = capi.Config(Precision=1e-6, Mu=0.05, IntensityDerivativeThreshold=2.0, DepthDerivativeThreshold=0.01)._c()
    Cn._lib().dvo_amd_default_validator_stages(C.byref(front), 0.25, 0.6, 0.8, cst)

    class F:
        precision, mu, intensity_derivative_threshold, depth_derivative_threshold = 1e-6, 0.05, 2.0, 0.01

    ref = V.create_constraint_proposal_validator(F, 0.25, 0.6, 0.8)
    names = {0: "OdometryConstraintVoter", 1: "NaNResultVoter", 2: "ConstraintRatioVoter", 3: "TrackingResultEvaluationVoter",
             4: "CrossValidationVoter"}
    for s, o in zip(cst, ref.stages):
        assert (s.id, bool(s.only_keep_best)) == (o.Id, o.OnlyKeepBest)
        assert [names[s.voters[k].kind] for k in range(s.n_voters)] == [type(x).__name__ for x in o.Voters]
        for k in range(s.n_voters):
            thr = getattr(o.Voters[k], "RatioThreshold", getattr(o.Voters[k], "TranslationThreshold", 0.0))
            assert s.voters[k].threshold == thr
        t, c = s.tracking_config, o.TrackingConfig
        assert (t.first_level, t.last_level, t.max_iterations_per_level, t.use_initial_estimate) == \
               (c.first_level, c.last_level, c.max_iterations_per_level, c.use_initial_estimate)
        assert (t.precision, t.mu) == (c.precision, c.mu)
        assert (t.intensity_derivative_threshold, t.depth_derivative_threshold) == \
               (c.intensity_derivative_threshold, c.depth_derivative_threshold)


def test_initial_proposals_of_the_c_abi_equal_the_oracle(V, synth):
    from dvo_slam_amd import constraints as Cn

    poses = [np.eye(4)] + synth.loop_closure_poses(3)
    ckf = (Cn.CKeyframe * 4)()
    for i, T in enumerate(poses):
        ckf[i].id, ckf[i].pose = 10 * i, Cn._colmajor(T)
    cand = (C.c_int * 3)(1, 2, 3)
    out = (Cn.CProposal * 6)()
    assert Cn._lib().dvo_amd_proposals_for_candidates(ckf, 0, 3, cand, out) == 0
    kfs = [V.Keyframe(10 * i, None, T, None) for i, T in enumerate(poses)]
    ref = V.proposals_for_candidates(kfs[0], kfs[1:])
    for c, o in zip(out, ref):
        assert (c.reference, c.current) == (0, kfs.index(o.Current))
        assert np.allclose(np.array(c.initial_transformation[:]).reshape(4, 4).T, o.InitialTransformation, atol=1e-14)
    assert Cn._lib().dvo_amd_proposals_for_candidates(ckf, 0, 3, None, out) != 0


def test_validate_rejects_bad_arguments_without_touching_the_gpu():
    from dvo_slam_amd import constraints as Cn

    n = C.c_int(7)
    assert Cn._lib().dvo_amd_validate_proposals(None, 0, None, 0, None, 0, None, C.byref(n), 0) != 0
    assert n.value == 0


# ---------------------------------------------------------------------------------------------------------------------
# GPU parity: the batched native validator against the oracle
# ---------------------------------------------------------------------------------------------------------------------
POSE_TOL = 1e-5  # BASELINE.json: ||log(T_ref^-1 T_gpu)|| <= 1e-5
import fork_criterion  # noqa: E402  (tests/fork_criterion.py: what a forked alignment may differ by)


def _summary_oracle(props):
    return [(p.Reference.id, p.Current.id, [v.Decision for v in p.Votes], [v.Value for v in p.Votes],
             p.TrackingResult["T"], p.TotalScore(), p.origin) for p in props]


def _summary_gpu(props):
    return [(p.Reference.id, p.Current.id, [v.Decision for v in p.Votes], [v.Value for v in p.Votes],
             p.TrackingResult.Transformation, p.TotalScore(), p.origin) for p in props]


_REMATCH = {}  # one tracker per stage configuration for the single match() that settles a fork


def _like_with_like(orc, synth, ov, stage, g_props, notes, sensor=False):
    """Every GPU survivor of `stage` against the oracle's alignment of the SAME proposal -- same (reference, current) and same
    initialisation lineage (`origin`), whether or not the oracle's own keepBest / cross-validation kept it: 1e-5.  A survivor
    beyond it is settled by the rule of tests/fork_criterion.py: the validator's batch returns no per-iteration statistics, so
    the same alignment is re-run through the single match() (which must give the batch's bits), the flipped decision is
    adjudicated and the oracle continued from the GPU's own state behind it.  ov.history holds every alignment the oracle
    validator ran."""
    from dvo_slam_amd import capi

    hist = {h[1]: h for h in ov.history if h[0] == stage.Id}
    worst = 0.0
    ocfg = stage.TrackingConfig
    for p in g_props:
        assert p.origin in hist, (p.origin, sorted(hist))
        _, _, rid, cid, init, ro = hist[p.origin]
        assert (rid, cid) == (p.Reference.id, p.Current.id)
        artefact = fork_criterion.overflow_note(ro)
        if artefact:
            notes.append(f"{rid}->{cid} origin {p.origin}: {artefact}")
        err = synth.pose_error(ro["T"], p.TrackingResult.Transformation)
        if err > POSE_TOL:
            key = tuple(getattr(ocfg, f) for f, _ in ocfg._fields_)
            if key not in _REMATCH:
                _REMATCH[key] = capi.DenseTracker(fork_criterion.gpu_config_of(capi, ocfg))
            # the very alignment the validator ran: the initial transformation it started this proposal from (the inverse proposals
            # of the cross-validation are formed inside the call; theirs differs from the oracle's numpy inverse in the last bit)
            assert np.allclose(p.stage_initial, init, rtol=0, atol=1e-13)
            rg = _REMATCH[key].match(p.Reference.image, p.Current.image, p.stage_initial)
            forked, report = fork_criterion.settle(orc, synth, ocfg, ov.images[rid], ov.images[cid], init, rg, ro, POSE_TOL,
                                                   batch_T=p.TrackingResult.Transformation, count_slack=40 if sensor else None,
                                                   increment_band=0.05 if sensor else 0.0)
            notes.append(f"{rid}->{cid} origin {p.origin}: {err:.2e} from the free-running oracle; " + " | ".join(report[1:] if forked else report))
        worst = max(worst, err)
    return worst


def _compare(orc, synth, ov, got_props, want_props, notes, sensor=False):
    """Survivors, their order, every vote decision; vote values within the heuristic band of a likelihood ratio.  Poses are
    compared like with like for a single stage only: in a free-running two-stage run each side starts stage 2 from its OWN
    stage-1 estimate (validator.cpp:95-100), so the two sides align different proposals there -- the second stage is compared
    teacher-forced instead (_teacher_forced_second_stage)."""
    got, want = _summary_gpu(got_props), _summary_oracle(want_props)
    assert [(g[0], g[1]) for g in got] == [(w[0], w[1]) for w in want]
    for g, w in zip(got, want):
        assert g[2] == w[2], (g[0], g[1], g[2], w[2], g[3], w[3])
        # vote values are likelihood ratios (and the norm of a pose difference): they move by ~1e-3 relative when the two sides
        # fork by one accepted step; thresholds are placed mid-gap so that decisions cannot flip on that
        assert np.allclose(g[3], w[3], rtol=1e-2, atol=6e-4), (g[3], w[3])
        assert abs(g[5] - w[5]) <= 1e-2 * max(1.0, abs(w[5]))
    if len(ov.stages) == 1:
        return _like_with_like(orc, synth, ov, ov.stages[0], got_props, notes, sensor)
    return 0.0


def _teacher_forced_second_stage(orc, synth, Cn, V, make_validators, o_stage1, gkf, okf, notes, sensor=False):
    """Stage 2 alone on both sides, both starting from the ORACLE's stage-1 survivors (reference, current, initial transformation
    = inverse of the oracle's stage-1 estimate): every GPU survivor against the oracle's alignment of the same proposal.
    make_validators() -> (GPU validator, oracle validator) holding the second stage only."""
    gv2, ov2 = make_validators()
    gp2 = [Cn.ConstraintProposal(gkf[q.Reference.id], gkf[q.Current.id], q.InitialTransformation) for q in o_stage1]
    op2 = [V.ConstraintProposal(okf[q.Reference.id], okf[q.Current.id], q.InitialTransformation) for q in o_stage1]
    g2, o2 = gv2.validate(gp2), ov2.validate(op2)
    assert len(ov2.history) == len(o_stage1)
    assert {frozenset((p.Reference.id, p.Current.id)) for p in g2} == {frozenset((p.Reference.id, p.Current.id)) for p in o2}
    return g2, o2, _like_with_like(orc, synth, ov2, ov2.stages[0], g2, notes, sensor)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["LogLikelihood", "NormalizedLogLikelihood", "EntropyRatio"])
def test_gpu_validator_equals_oracle(orc, V, synth, kind):
    _validator_against_oracle(orc, V, synth, kind, sensor=False)


@pytest.mark.gpu
def test_gpu_validator_equals_oracle_on_sensor_frames(orc, V, synth):
    """the same scenario as the sensor delivers it: 8-bit grey, uint16 depth at 1/5000 m with the depth noise the reference models
    (synth.sensor_frame) -- the regime the reference's validator actually runs in"""
    _validator_against_oracle(orc, V, synth, "LogLikelihood", sensor=True)


def _validator_against_oracle(orc, V, synth, kind, sensor):
    from dvo_slam_amd import capi, constraints as Cn

    if capi.lib().dvo_amd_device_count() < 1:
        pytest.fail("no HIP device visible: the gpu tests must run on the MI355X box")
    n_cand = 6
    okey, ocands = S.oracle_keyframes(orc, V, synth, 640, 480, n_cand, getattr(V, kind + "TrackingResultEvaluation"), sensor=sensor)
    gkey, gcands = S.gpu_keyframes(capi, Cn, synth, 640, 480, n_cand, getattr(Cn, kind + "TrackingResultEvaluation"), sensor=sensor)
    assert abs(gkey.evaluation.average - okey.evaluation.average) <= 1e-4 * abs(okey.evaluation.average)

    notes = []
    images = {k.id: k.image for k in [okey] + ocands}

    def run(thresholds, stages=2):
        ov = V.create_constraint_proposal_validator(**thresholds)
        gv = Cn.createConstraintProposalValidator(**thresholds)
        ov.stages, gv.stages = ov.stages[:stages], gv.stages[:stages]
        ov.images = images
        o = ov.validate(V.proposals_for_candidates(okey, ocands))
        g = gv.validate(Cn.proposalsForCandidates(gkey, gcands))
        worst = _compare(orc, synth, ov, g, o, notes, sensor)
        return o, worst

    # (1) stage 1 alone, nothing rejected by a ratio: the observed coarse ratios give a threshold that splits the proposals
    o1, _ = run(PERMISSIVE, stages=1)
    assert len(o1) == 2 * (n_cand + 1)  # keepAll: both initialisations of every pair; the two hard decoys are gone
    coarse = S.mid_gap_threshold([p.Votes[3].Value for p in o1])
    # (2) both stages with that coarse threshold; the observed fine ratios give the fine threshold
    o2, _ = run(dict(PERMISSIVE, ratio_coarse=coarse))
    assert 0 < len(o2) < len(o1)
    fine = S.mid_gap_threshold([p.Votes[2].Value for p in o2])
    # (3) the full decision chain, and with a constraint-ratio floor that bites
    o3, worst = run(dict(min_constraint_ratio=0.0, ratio_coarse=coarse, ratio_fine=fine))
    assert 0 < len(o3) < len(o2) or len(o2) == 1
    ratios = [p.Votes[1].Value for p in o2]
    run(dict(min_constraint_ratio=S.mid_gap_threshold(ratios), ratio_coarse=coarse, ratio_fine=-1e300))
    # (4) the reference's default thresholds (config.cpp:38-43)
    run(dict(min_constraint_ratio=0.2, ratio_coarse=0.7, ratio_fine=0.9))
    # (5) the second stage like with like: both sides start it from the oracle's stage-1 survivors
    gkf = {k.id: k for k in [gkey] + gcands}
    okf = {k.id: k for k in [okey] + ocands}

    def second_stage_only():
        ov = V.create_constraint_proposal_validator(**PERMISSIVE)
        gv = Cn.createConstraintProposalValidator(**PERMISSIVE)
        ov.stages, gv.stages = ov.stages[1:], gv.stages[1:]
        ov.images = images
        return gv, ov
    _, _, worst2 = _teacher_forced_second_stage(orc, synth, Cn, V, second_stage_only, o1, gkf, okf, notes, sensor)
    regime = "sensor regime" if sensor else "analytic regime"
    print(f"[validator {kind}, {regime}] second stage, teacher-forced: worst pose error vs the oracle's alignment of the same proposal "
          f"{worst2:.2e}; {len(notes)} alignments beyond 1e-5 of the free-running oracle (each forked, adjudicated and within 1e-5 of the "
          f"oracle continued from the GPU's own state behind the fork)")
    for n in notes:
        print(f"[validator fork, {regime}]", n)


@pytest.mark.gpu
def test_config5_full_size_32_candidates_against_the_oracle(orc, V, synth):
    """BASELINE config 5 at its stated size: one keyframe against 32 candidate frames, two proposals each = 64 proposals,
    both stages, every alignment FirstLevel 3 -> LastLevel 0 (the metric's configuration), decisions and poses against
    oracle/validator.py (the oracle side is ~200 CPU alignments)."""
    from dvo_slam_amd import capi, constraints as Cn

    n_cand = 32
    key, cands = synth.loop_closure_scenario(640, 480, n_cand, decoys=False)
    K = synth.intrinsics_for(640, 480)
    cfg = dict(FirstLevel=3, LastLevel=0)
    ocfg = orc.default_config(first_level=3, last_level=0, rcp_mode=orc.RCP_EXACT)
    trk = capi.DenseTracker(capi.Config(**cfg))

    def both(e):
        gp, op = capi.RgbdImagePyramid(e["frame"][0], e["frame"][1], K, 4), orc.Pyramid(e["frame"][0], e["frame"][1], K, 4)
        nb = S.neighbour_frame(synth, e, 640, 480)
        gk = Cn.Keyframe(e["id"], gp, e["pose"], Cn.LogLikelihoodTrackingResultEvaluation(trk.match(gp, capi.RgbdImagePyramid(nb[0], nb[1], K, 4))))
        ok = V.Keyframe(e["id"], op, e["pose"], V.LogLikelihoodTrackingResultEvaluation(orc.match(ocfg, op, orc.Pyramid(nb[0], nb[1], K, 4))))
        return gk, ok

    gkey, okey = both(key)
    pairs = [both(c) for c in cands]
    gcands, ocands = [p[0] for p in pairs], [p[1] for p in pairs]
    gkf = {k.id: k for k in [gkey] + gcands}
    okf = {k.id: k for k in [okey] + ocands}
    images = {k.id: k.image for k in [okey] + ocands}
    th = dict(min_constraint_ratio=0.2, ratio_coarse=-1e300, ratio_fine=-1e300)

    def validators(first_stage, n_stages):
        gv = Cn.createConstraintProposalValidator(tracker=trk, max_in_flight=72, **th)
        ov = V.create_constraint_proposal_validator(**th)
        for st in gv.stages:  # "all First=3, Last=0 for the metric" 
            st.TrackingConfig.FirstLevel, st.TrackingConfig.LastLevel = 3, 0
        for st in ov.stages:
            st.TrackingConfig.first_level, st.TrackingConfig.last_level = 3, 0
        gv.stages, ov.stages = gv.stages[first_stage:first_stage + n_stages], ov.stages[first_stage:first_stage + n_stages]
        ov.images = images
        return gv, ov

    notes = []
    # ---- (1) stage 1 alone: 64 proposals + 64 cross-validation inverses, every GPU survivor against the oracle's alignment of
    # the SAME (reference, current, initial transformation) -- the oracle ran all 128 -- under the 1e-5 / fork rule
    gv1, ov1 = validators(0, 1)
    g1 = gv1.validate(Cn.proposalsForCandidates(gkey, gcands))
    o1 = ov1.validate(V.proposals_for_candidates(okey, ocands))
    assert len(ov1.history) == 128 and len(g1) == len(o1) == 64
    worst1 = _like_with_like(orc, synth, ov1, ov1.stages[0], g1, notes)
    # ---- (2) stage 2 alone, teacher-forced: both sides start from the ORACLE's stage-1 survivors (stage 2 starts from stage 1's
    # estimate, validator.cpp:95-100: in a free-running run each side would start from its own), keepBest, like with like
    g2, o2, worst2 = _teacher_forced_second_stage(orc, synth, Cn, V, lambda: validators(1, 1), o1, gkf, okf, notes)
    assert len(g2) == len(o2) == n_cand
    same_survivor = sum(p.origin == q.origin for p, q in zip(g2, o2))
    print(f"[config 5] like with like: stage 1 worst {worst1:.2e} over {len(g1)} survivors, stage 2 (teacher-forced) worst "
          f"{worst2:.2e} over {len(g2)}; keepBest picked the same initialisation on both sides for {same_survivor} of {n_cand}; "
          f"{len(notes)} alignments beyond 1e-5 of the free-running oracle, each forked, adjudicated and re-synchronised:")
    for n in notes:
        print("    [validator fork]", n)
    # ---- (3) end to end, free running: same pairs survive with the same vote decisions
    gv, ov = validators(0, 2)
    g, o = gv.validate(Cn.proposalsForCandidates(gkey, gcands)), ov.validate(V.proposals_for_candidates(okey, ocands))
    assert len(g) == len(o) == n_cand  # keepBest: one constraint per candidate survives
    # keepBest chooses between a proposal and its cross-validation inverse by a likelihood-ratio score; the two scores of a
    # pair are within ~1 % of each other and GPU / oracle likelihoods differ by that much when their last iterations differ,
    # so the surviving DIRECTION (and initialisation) may differ: decisions are compared per unordered pair, poses like with
    # like in (1) and (2)
    og = {frozenset((p.Reference.id, p.Current.id)): p for p in o}
    assert {frozenset((p.Reference.id, p.Current.id)) for p in g} == set(og)
    for p in g:
        q = og[frozenset((p.Reference.id, p.Current.id))]
        assert [v.Decision for v in p.Votes] == [v.Decision for v in q.Votes]
    truth = {c["id"]: c["pose_true"] for c in cands}
    truth[key["id"]] = np.eye(4)
    for p in g:  # and every kept constraint is the right relative pose (cur <- ref, dense_tracking.cpp:371)
        assert not p.TrackingResult.isNaN()
        want = truth[p.Current.id] @ np.linalg.inv(truth[p.Reference.id])
        assert synth.pose_error(want, p.TrackingResult.Transformation) < 2e-3


@pytest.mark.gpu
def test_gpu_validator_in_flight_limit_and_config_restore(synth):
    from dvo_slam_amd import capi, constraints as Cn

    gkey, gcands = S.gpu_keyframes(capi, Cn, synth, 640, 480, 4)
    trk = capi.DenseTracker(capi.Config(FirstLevel=2, LastLevel=1, MaxIterationsPerLevel=17))
    a = Cn.createConstraintProposalValidator(tracker=trk, **PERMISSIVE).validate(Cn.proposalsForCandidates(gkey, gcands))
    c = capi.CConfig()
    assert capi.lib().dvo_amd_get_config(trk._h, C.byref(c)) == 0
    assert (c.first_level, c.last_level, c.max_iterations_per_level) == (2, 1, 17)  # the caller's configuration is back
    b = Cn.createConstraintProposalValidator(max_in_flight=3, **PERMISSIVE).validate(Cn.proposalsForCandidates(gkey, gcands))
    assert [(p.Reference.id, p.Current.id) for p in a] == [(p.Reference.id, p.Current.id) for p in b]
    # 3 resident pairs or all of them: what a pair shares its ticks with does not enter its result (tests/test_determinism.py)
    for p, q in zip(a, b):
        assert np.array_equal(p.TrackingResult.Transformation, q.TrackingResult.Transformation)
        assert np.array_equal(p.TrackingResult.Information, q.TrackingResult.Information)
    assert Cn.createConstraintProposalValidator(**PERMISSIVE).validate([]) == []


@pytest.mark.gpu
def test_validator_output_does_not_depend_on_the_worker_count(synth, monkeypatch):
    """dvo_amd_validate_proposals deals a stage of at least 48 alignments over up to DVO_AMD_VALIDATOR_THREADS worker contexts
    (the reference deals its proposals over TBB workers, keyframe_graph.cpp:587-590): 1, 2 or 3 workers, 72 or 5 resident
    pairs -- the surviving constraints, their votes and their poses are the same bits (until round 3 a worker's share decided the
    wave-segment length of its ticks and with it the last bits of every sum)."""
    from dvo_slam_amd import capi, constraints as Cn

    gkey, gcands = S.gpu_keyframes(capi, Cn, synth, 640, 480, 30)  # 30 + 3 decoys: 66 proposals, 132 alignments in stage 1
    runs = []
    for threads, in_flight in (("1", 72), ("3", 72), ("2", 5), ("3", 0)):
        monkeypatch.setenv("DVO_AMD_VALIDATOR_THREADS", threads)
        v = Cn.createConstraintProposalValidator(max_in_flight=in_flight, **PERMISSIVE)
        runs.append(v.validate(Cn.proposalsForCandidates(gkey, gcands)))
    assert len(runs[0]) >= 25
    for other in runs[1:]:
        assert [(p.Reference.id, p.Current.id, p.origin) for p in runs[0]] == [(p.Reference.id, p.Current.id, p.origin) for p in other]
        for p, q in zip(runs[0], other):
            assert np.array_equal(p.TrackingResult.Transformation, q.TrackingResult.Transformation)
            assert np.array_equal(p.TrackingResult.Information, q.TrackingResult.Information)
            assert p.TrackingResult.LogLikelihood == q.TrackingResult.LogLikelihood
            assert [(v.Decision, v.Score) for v in p.Votes] == [(v.Decision, v.Score) for v in q.Votes]
