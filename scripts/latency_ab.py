"""Single-pair latency A/B: mean match() time of the headline pair and of 10 bench pairs, levels 3..0 and 3..1, plus the
two-pair front-end step.  usage: [DVO_AMD_WAITER=0] python scripts/latency_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dvo_slam_amd import capi, synth
K = synth.intrinsics_for(640, 480)
(Ir, Zr), (Ic, Zc), _ = synth.make_pair(640, 480)
ref, cur = capi.RgbdImagePyramid(Ir, Zr, K, 4), capi.RgbdImagePyramid(Ic, Zc, K, 4)
others = [capi.RgbdImagePyramid(*synth.render(640, 480, synth.se3_exp(synth.XI_GT_PAIR * (0.5 + 0.1 * i)), frame_id=3 + 2 * i), K, 4) for i in range(6)]
for name, cfg in (("levels 3..0", capi.Config(FirstLevel=3, LastLevel=0)), ("levels 3..1", capi.Config(FirstLevel=3, LastLevel=1))):
    trk = capi.DenseTracker(cfg)
    for _ in range(20):
        r = trk.match(ref, cur)
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(50):
            r = trk.match(ref, cur)
        best = min(best, (time.perf_counter() - t0) / 50)
    t0 = time.perf_counter()
    ticks = 0
    for i in range(60):
        ticks += trk.match(ref, others[i % 6]).n_ticks
    mean6 = (time.perf_counter() - t0) / 60
    print(f"{name}: headline pair {best * 1e3:.4f} ms ({r.n_ticks} ticks, {best * 1e6 / r.n_ticks:.2f} us per tick); 6 other pairs mean {mean6 * 1e3:.4f} ms ({mean6 * 1e6 * 60 / ticks:.2f} us per tick)", flush=True)
trk = capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=1, UseInitialEstimate=True))
eye = np.eye(4)
for _ in range(10):
    trk.track_frame(ref, others[0], others[1], eye)
t0 = time.perf_counter()
for i in range(40):
    trk.track_frame(ref, others[i % 5], others[i % 5 + 1], eye)
print(f"two-pair front-end step: {(time.perf_counter() - t0) / 40 * 1e3:.4f} ms", flush=True)
