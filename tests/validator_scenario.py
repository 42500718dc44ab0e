"""Shared set-up of the loop-closure validation tests: the same scenario for the oracle (CPU) and the HIP path (GPU)."""
import numpy as np

LEVELS = 4


def neighbour_frame(synth, entry, width, height, sensor=False):
    """A frame one odometry step away from the keyframe: its alignment gives the keyframe's TrackingResultEvaluation, the
    way KeyframeTracker seeds it from the first odometry result (keyframe_tracker.cpp:88-96).  sensor: the frame as the sensor
    delivers it (synth.sensor_frame), ingested to float planes."""
    T = synth.se3_exp(synth.XI_STEP_STREAM) @ entry["pose_true"]
    seed = synth.SEED + 77 if entry["id"] == 60 else synth.SEED
    if sensor:
        return synth.raw_to_float(*synth.sensor_frame(width, height, T, seed, 200 + entry["id"]))
    return synth.render(width, height, T, seed, 200 + entry["id"])


def oracle_keyframes(orc, V, synth, width, height, n_candidates, evaluation_cls=None, sensor=False):
    evaluation_cls = evaluation_cls or V.LogLikelihoodTrackingResultEvaluation
    K = synth.intrinsics_for(width, height)
    key, cands = synth.loop_closure_scenario(width, height, n_candidates, sensor=sensor)
    cfg = orc.default_config(first_level=3, last_level=1, rcp_mode=orc.RCP_EXACT)

    def mk(e):
        p = orc.Pyramid(e["frame"][0], e["frame"][1], K, LEVELS)
        nb = neighbour_frame(synth, e, width, height, sensor)
        r = orc.match(cfg, p, orc.Pyramid(nb[0], nb[1], K, LEVELS))
        kf = V.Keyframe(e["id"], p, e["pose"], evaluation_cls(r))
        kf.pose_true = e["pose_true"]
        return kf

    return mk(key), [mk(c) for c in cands]


def gpu_keyframes(capi, Cn, synth, width, height, n_candidates, evaluation_cls=None, sensor=False):
    evaluation_cls = evaluation_cls or Cn.LogLikelihoodTrackingResultEvaluation
    K = synth.intrinsics_for(width, height)
    key, cands = synth.loop_closure_scenario(width, height, n_candidates, sensor=sensor)
    trk = capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=1))

    def mk(e):
        p = capi.RgbdImagePyramid(e["frame"][0], e["frame"][1], K, LEVELS)
        nb = neighbour_frame(synth, e, width, height, sensor)
        r = trk.match(p, capi.RgbdImagePyramid(nb[0], nb[1], K, LEVELS))
        kf = Cn.Keyframe(e["id"], p, e["pose"], evaluation_cls(r))
        kf.pose_true = e["pose_true"]
        return kf

    return mk(key), [mk(c) for c in cands]


def mid_gap_threshold(values):
    """A threshold in the middle of the widest gap of the observed values: splits them robustly into accept / reject."""
    v = np.sort(np.asarray([x for x in values if np.isfinite(x)], dtype=np.float64))
    if len(v) < 2:
        return -1e300
    gaps = np.diff(v)
    k = int(np.argmax(gaps))
    return float(0.5 * (v[k] + v[k + 1]))
