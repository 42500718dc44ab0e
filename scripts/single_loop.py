"""Runs N single-pair matches and (optionally) M batch matches; used under rocprofv3 to look at the tick timeline."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvo_slam_amd import synth, capi
n_single = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n_batch = int(sys.argv[2]) if len(sys.argv) > 2 else 0
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
w, h = 640, 480
K = synth.intrinsics_for(w, h)
(Ir, Zr), (Ic, Zc), Tgt = synth.make_pair(w, h)
pr, pc = capi.RgbdImagePyramid(Ir, Zr, K, 4), capi.RgbdImagePyramid(Ic, Zc, K, 4)
trk = capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=0))
trk.match(pr, pc)
t = time.perf_counter()
for i in range(n_single):
    r = trk.match(pr, pc)
dt = (time.perf_counter() - t) / max(n_single, 1)
print("single ms", dt * 1e3, "ticks", r.n_ticks)
if n_batch:
    trk.match_batch([pr] * B, [pc] * B, stats=False)
    t = time.perf_counter()
    for i in range(n_batch):
        trk.match_batch([pr] * B, [pc] * B, stats=False)
    dt = (time.perf_counter() - t) / n_batch
    print("batch ms", dt * 1e3, "pairs/s", B / dt)
