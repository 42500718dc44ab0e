"""Time of dvo_amd_validate_proposals on BASELINE config 5 (64 proposals) against the pairs kept in flight."""
import sys
import time

sys.path.insert(0, ".")
from dvo_slam_amd import capi, constraints as Cn, synth

W, H = 640, 480
K = synth.intrinsics_for(W, H)
key, cands = synth.loop_closure_scenario(W, H, 32, decoys=False)
trk = capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=1))


def mk(e):
    p = capi.RgbdImagePyramid(e["frame"][0], e["frame"][1], K, 4)
    return Cn.Keyframe(e["id"], p, e["pose"], Cn.LogLikelihoodTrackingResultEvaluation(trk.match(p, p)))


kkey, kc = mk(key), [mk(c) for c in cands]
for in_flight in (36, 72, 144, 0):
    val = Cn.createConstraintProposalValidator(min_constraint_ratio=0.2, ratio_coarse=-1e300, ratio_fine=-1e300,
                                               max_in_flight=in_flight)
    val.validate(Cn.proposalsForCandidates(kkey, kc))
    t0 = time.perf_counter()
    for _ in range(10):
        props = Cn.proposalsForCandidates(kkey, kc)
        t1 = time.perf_counter()
        out = val.validate(props)
    dt = (time.perf_counter() - t0) / 10
    print(f"in flight {in_flight:3d}: {dt * 1e3:.2f} ms per validate (+ proposals), {len(out)} constraints kept")
