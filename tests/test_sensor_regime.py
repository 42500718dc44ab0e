"""The parity suite once more in the regime the reference actually runs in.

dvo_benchmark feeds 8-bit grey and uint16 depth at 1/5000 m (dvo_benchmark/src/benchmark_slam.cpp:56-80) from a sensor whose depth
noise the reference models itself, depthStdDevZ(z) = 0.0012 + 0.0019 (z - 0.4)^2 (dvo_core/src/dense_tracking_impl.cpp:122-128).
Every other GPU test of this repository runs on NOISE-FREE analytic depth, where the depth precision comes out as 1e9, the
reference's 50-term likelihood product overflows and one valid pixel more or less flips an iteration.  Here the same checks run
on synth.sensor_frame() input -- hashed Gaussian depth noise of exactly that sigma, 0.2 mm quantisation, 8-bit grey with sensor
noise -- that enters through dvo_amd_pyramid_create_raw (device-side ingest) on the GPU side and orc_ingest_* on the oracle's:

  bit-exact:       pyramid planes, point selection, residuals + validity at every level and four poses, error image
  teacher-forced:  every iteration of full oracle matches replayed on the GPU (640x480 both ways, 1280x960)
  free-running:    match() against the oracle for eleven configurations, every fork adjudicated (tests/fork_criterion.py)
  next to the path: the dual-match front-end step, the loop-closure validator (tests/test_validator.py::..._on_sensor_frames)

The last test prints the book of this regime next to the noise-free one's: forks, configurations beyond 1e-5, worst pose error.
PARITY UNPINNED like everything else here: the oracle is this repository's restatement of the reference.
"""
import numpy as np
import pytest

import test_gpu_parity as P

pytestmark = pytest.mark.gpu

SENSOR_PATHS = {"same": [], "forked": [], "fork_err": [], "reports": [], "errs": []}


@pytest.fixture(scope="module")
def capi():
    from dvo_slam_amd import capi as c

    if c.lib().dvo_amd_device_count() < 1:
        pytest.fail("no HIP device visible: the gpu tests must run on the MI355X box")
    return c


def _both(capi, orc, raw, K, levels):
    """the raw frame through the device-side ingest (dvo_amd_pyramid_create_raw) and through the oracle's restatement of it"""
    gray, raw_z = raw
    return (capi.RgbdImagePyramid.from_raw(gray, raw_z, K, levels),
            orc.Pyramid(orc.ingest_gray(gray), orc.ingest_depth(raw_z), K, levels))


def _sensor_pair(capi, orc, synth, w, h, levels, xi=None, **kw):
    ref, cur, Tgt = synth.sensor_pair(w, h, **({} if xi is None else {"xi_gt": xi}), **kw)
    K = synth.intrinsics_for(w, h)
    gr, orr = _both(capi, orc, ref, K, levels)
    gc, occ = _both(capi, orc, cur, K, levels)
    return dict(gr=gr, gc=gc, orr=orr, occ=occ, Tgt=Tgt, K=K, raw=(ref, cur),
                frames=(synth.raw_to_float(*ref), synth.raw_to_float(*cur)))


@pytest.fixture(scope="module")
def spair(capi, orc, synth):
    return _sensor_pair(capi, orc, synth, 640, 480, 4)


# ---------------------------------------------------------------------------------------------------------------------
# bit-exact stages (the functions of tests/test_gpu_parity.py, fed the sensor pair)
# ---------------------------------------------------------------------------------------------------------------------
def test_ingested_planes_and_pyramid_bit_exact(spair):
    P.test_pyramid_planes_bit_exact(spair)
    # and the float planes the CPU-side helper makes of the raw frame are the ingested ones (tests that take float frames --
    # the validator, the front end -- feed those)
    for which, k in (("gr", 0), ("gc", 1)):
        If, Zf = spair["frames"][k]
        assert np.array_equal(spair[which].plane(0, 0), If)
        assert np.array_equal(spair[which].plane(0, 1), Zf, equal_nan=True)


@pytest.mark.parametrize("thresholds", [(0.0, 0.0), (2.5, 0.01)])
def test_point_selection_identical(spair, thresholds):
    P.test_point_selection_identical(spair, thresholds)


@pytest.mark.parametrize("level", [3, 2, 1, 0])
def test_residuals_and_validity_bit_exact(capi, orc, synth, spair, level):
    P.test_residuals_and_validity_bit_exact(capi, orc, synth, spair, level)


def test_error_image_bit_exact(capi, orc, spair):
    P.test_error_image_matches_the_oracle(capi, orc, spair)


@pytest.mark.parametrize("level", [3, 2, 1, 0])
def test_weighted_iteration_stages(capi, orc, synth, spair, level):
    P.test_weighted_iteration_stages_match_the_oracle(capi, orc, synth, spair, level)


# ---------------------------------------------------------------------------------------------------------------------
# every iteration, teacher-forced
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["640x480 sensor", "640x480 sensor swapped", "1280x960 sensor"])
def test_every_iteration_teacher_forced(capi, orc, synth, spair, case, capsys):
    if case.startswith("1280x960"):
        big = _sensor_pair(capi, orc, synth, 1280, 960, 5)
        worst = P._teacher_forced(capi, orc, synth, big["gr"], big["gc"], big["orr"], big["occ"], 4, case, capsys)
    elif "swapped" in case:
        worst = P._teacher_forced(capi, orc, synth, spair["gc"], spair["gr"], spair["occ"], spair["orr"], 3, case, capsys)
    else:
        worst = P._teacher_forced(capi, orc, synth, spair["gr"], spair["gc"], spair["orr"], spair["occ"], 3, case, capsys)
    assert worst["ref_ll"] <= P.REF_LL_RTOL


# ---------------------------------------------------------------------------------------------------------------------
# free-running matches against the oracle, forks adjudicated one by one
# ---------------------------------------------------------------------------------------------------------------------
def _check(capi, orc, synth, d, cfg_kw, T_init=None, swap=False):
    import inspect

    gr, gc, orr, occ = (d["gc"], d["gr"], d["occ"], d["orr"]) if swap else (d["gr"], d["gc"], d["orr"], d["occ"])
    label = "sensor " + inspect.stack()[1].function + repr(sorted(cfg_kw.items()))
    # (drift band of the per-iteration comparison: on sensor input the two sides' poses are ~1e-7 .. 1e-6 apart at the late
    #  iterations, and the 2 % of missing depth readings give a level some 50 000 cell edges a projected point can sit next to:
    #  a handful to a few dozen valid-constraint flips, against 0 .. 3 on the noise-free frames)
    # (and the same drift moves an increment by a few per cent of its largest component: increment_band, this regime only)
    return P._check_match(capi, orc, synth, gr, gc, orr, occ, cfg_kw, T_init, paths=SENSOR_PATHS, label=label, count_slack=40,
                          increment_band=0.05)


def test_match_640x480_4_levels(capi, orc, synth, spair):
    rg, ro, err = _check(capi, orc, synth, spair, dict(FirstLevel=3, LastLevel=0))
    # the right answer to what the sensor's noise allows: sigma_z is 3 mm at the scene's depths, 300 000 pixels
    assert synth.pose_error(spair["Tgt"], rg.Transformation) < 1e-3
    # depth precision of sensor data: ~1e4 .. 1e5 (the noise-free frames: 1e9), nowhere near the likelihood's overflow
    P_last = rg.Levels[-1]["Iterations"][-1]["TDistributionPrecision"]
    assert 1e3 < P_last[1, 1] < 1e6, P_last
    assert not any(np.isinf(it["TDistributionLogLikelihood"]) for L in rg.Levels for it in L["Iterations"])


def test_match_reference_default_levels(capi, orc, synth, spair):
    _check(capi, orc, synth, spair, dict(FirstLevel=3, LastLevel=1))


def test_match_swapped_roles(capi, orc, synth, spair):
    _check(capi, orc, synth, spair, dict(FirstLevel=3, LastLevel=0), swap=True)


def test_match_with_initial_estimate_and_prior(capi, orc, synth, spair):
    T0 = synth.se3_exp(synth.XI_GT_PAIR * 0.8)
    _check(capi, orc, synth, spair, dict(FirstLevel=3, LastLevel=1, UseInitialEstimate=True, Mu=0.05), T_init=T0)
    _check(capi, orc, synth, spair, dict(FirstLevel=2, LastLevel=0, UseInitialEstimate=True), T_init=T0)


def test_match_with_gradient_thresholds(capi, orc, synth, spair):
    _check(capi, orc, synth, spair, dict(FirstLevel=3, LastLevel=0, IntensityDerivativeThreshold=2.5, DepthDerivativeThreshold=0.01))


def test_match_larger_motion_and_iteration_cap(capi, orc, synth):
    d = _sensor_pair(capi, orc, synth, 640, 480, 4, xi=synth.XI_GT_PAIR * 2.5, frame_id=3)
    _check(capi, orc, synth, d, dict(FirstLevel=3, LastLevel=0))
    _check(capi, orc, synth, d, dict(FirstLevel=3, LastLevel=0, MaxIterationsPerLevel=3))


@pytest.mark.parametrize("size", [(336, 250, 3), (1280, 960, 5)])
def test_match_other_sizes(capi, orc, synth, size):
    w, h, levels = size
    d = _sensor_pair(capi, orc, synth, w, h, levels, xi=synth.XI_GT_PAIR * (0.4 if w < 640 else 1.0))
    _check(capi, orc, synth, d, dict(FirstLevel=levels - 1, LastLevel=0))


def test_batched_sensor_pairs_equal_their_single_matches(capi, orc, synth, spair):
    """the determinism property of tests/test_determinism.py on sensor input (raw ingest included)"""
    trk = capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=0))
    a, b = trk.match(spair["gr"], spair["gc"]), trk.match(spair["gc"], spair["gr"])
    out = trk.match_batch([spair["gr"], spair["gc"]] * 30, [spair["gc"], spair["gr"]] * 30, in_flight=36)
    for k, r in enumerate(out):
        want = a if k % 2 == 0 else b
        assert np.array_equal(want.Transformation, r.Transformation) and np.array_equal(want.Information, r.Information)
    for n_bands in (2, 8):
        assert np.array_equal(trk.match_banded(spair["gr"], spair["gc"], n_bands).Transformation, a.Transformation)


# ---------------------------------------------------------------------------------------------------------------------
# next to the path
# ---------------------------------------------------------------------------------------------------------------------
def test_track_frame_on_sensor_frames(capi, orc, synth):
    def render(w, h, T, frame_id=0):
        return synth.raw_to_float(*synth.sensor_frame(w, h, T, frame_id=frame_id))

    P._track_frame_case(capi, orc, synth, render, SENSOR_PATHS, gt_tol=3e-3, count_slack=40, increment_band=0.05)


def test_zz_book_of_the_sensor_regime(capsys):
    """forks, configurations beyond 1e-5 and the worst pose error of this regime; the noise-free regime prints the same line
    (tests/test_gpu_parity.py::test_zz_every_fork_was_adjudicated)"""
    S = SENSOR_PATHS
    with capsys.disabled():
        print(f"\n[paths] sensor regime (8-bit grey, uint16 depth at 1/5000 m, depth noise sigma_z(z)): same path: {len(S['same'])}, "
              f"forked: {len(S['forked'])}; configurations beyond 1e-5 of the oracle: {sum(e > 1e-5 for e in S['errs'])} of "
              f"{len(S['errs'])}, worst {max(S['errs'] or [0.0]):.2e}; pose errors of the forked configurations: "
              f"{['%.1e' % e for e in S['fork_err']]}")
        print("[re-syncs per forked configuration, sensor regime] "
              + "; ".join(f"{label}: {sum(ln.startswith('re-sync') for ln in report)}" for label, report in S["reports"]))
        for label, report in S["reports"]:
            print(f"[fork, sensor regime] {label}")
            for line in report:
                print(f"        {line}")
    if not S["errs"]:
        pytest.skip("no match configuration of the sensor regime ran in this session")
    assert len(S["reports"]) == len(S["forked"])
