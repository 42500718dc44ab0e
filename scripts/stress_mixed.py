"""Mixed stress: several threads building pyramids from raw frames, running the validator, the front-end step and batched
matches at the same time on one GPU for a fixed budget; checks for crashes, NaNs and pose drift."""
import sys
import threading
import time

import numpy as np

sys.path.insert(0, ".")
from dvo_slam_amd import capi, constraints as Cn, synth

BUDGET = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
W, H = 640, 480
K = synth.intrinsics_for(W, H)
poses = synth.stream_poses(6)
frames = [synth.render(W, H, poses[t], frame_id=t) for t in range(6)]
raws = [synth.to_raw(I, Z) for I, Z in frames]
errors = []
counts = {"ingest": 0, "validate": 0, "track": 0, "batch": 0}
t_end = time.time() + BUDGET


def guard(fn):
    def run():
        try:
            fn()
        except Exception as exc:  # noqa: BLE001
            errors.append(repr(exc))
    return run


def ingest():
    while time.time() < t_end and not errors:
        i = counts["ingest"] % 6
        p = capi.RgbdImagePyramid.from_raw(raws[i][0], raws[i][1], K, 4)
        assert p.levels() == 4
        counts["ingest"] += 1


def validate():
    trk = capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=1))
    pyr = [capi.RgbdImagePyramid(I, Z, K, 4) for I, Z in frames]
    kfs = [Cn.Keyframe(10 * t, pyr[t], poses[t], Cn.LogLikelihoodTrackingResultEvaluation(trk.match(pyr[t], pyr[t]))) for t in range(6)]
    val = Cn.createConstraintProposalValidator(min_constraint_ratio=0.2, ratio_coarse=-1e300, ratio_fine=-1e300, max_in_flight=72)
    while time.time() < t_end and not errors:
        out = val.validate(Cn.proposalsForCandidates(kfs[0], kfs[1:]))
        assert len(out) == 5
        for p in out:
            want = np.asarray(p.Current.pose) @ np.linalg.inv(p.Reference.pose)
            assert synth.pose_error(p.TrackingResult.Transformation, want) < 2e-3
        counts["validate"] += 1


def track():
    trk = capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=1, UseInitialEstimate=True))
    pyr = [capi.RgbdImagePyramid(I, Z, K, 4) for I, Z in frames]
    while time.time() < t_end and not errors:
        t = 2 + counts["track"] % 4
        rk, ro, crit = trk.track_frame(pyr[0], pyr[t - 1], pyr[t], poses[t - 1])
        assert not rk.isNaN() and not ro.isNaN()
        assert synth.pose_error(rk.Transformation, poses[t]) < 2e-3
        counts["track"] += 1


def batch():
    trk = capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=0))
    pyr = [capi.RgbdImagePyramid(I, Z, K, 4) for I, Z in frames]
    while time.time() < t_end and not errors:
        out = trk.match_batch(pyr[:-1] * 20, pyr[1:] * 20, stats=False, in_flight=72, raw=True)
        assert not any(o.is_nan for o in out)
        counts["batch"] += 1


threads = [threading.Thread(target=guard(f)) for f in (ingest, validate, track, batch, batch)]
for t in threads:
    t.start()
for t in threads:
    t.join()
print(counts, "errors:", errors[:3])
sys.exit(1 if errors else 0)
