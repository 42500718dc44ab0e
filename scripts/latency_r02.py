"""Single-pair latency of match() (headline pair, levels 3..0) and of a 2-pair / 8-pair batch, after a GPU warm-up."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dvo_slam_amd import capi, synth
W, H = 640, 480
K = synth.intrinsics_for(W, H)
(Ir, Zr), (Ic, Zc), Tgt = synth.make_pair(W, H)
ref, cur = capi.RgbdImagePyramid(Ir, Zr, K, 4), capi.RgbdImagePyramid(Ic, Zc, K, 4)
trk = capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=0))
for _ in range(200):  # warm-up: clocks, pools
    r = trk.match(ref, cur)
def timeit(fn, n):
    t0 = time.perf_counter()
    for _ in range(n): out = fn()
    return (time.perf_counter() - t0) * 1e3 / n, out
ms1, r = timeit(lambda: trk.match(ref, cur), 300)
ms8, _ = timeit(lambda: trk.match_batch([ref] * 8, [cur] * 8, stats=False), 100)
t31 = capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=1))
for _ in range(50): t31.match(ref, cur)
ms31, r31 = timeit(lambda: t31.match(ref, cur), 300)
print(f"steps_at={os.environ.get('DVO_AMD_STEPS_AT','default')}: single pair {ms1:.4f} ms ({r.n_ticks} ticks, {ms1*1e3/r.n_ticks:.1f} us/tick), "
      f"8-pair batch {ms8:.4f} ms, levels 3..1 single pair {ms31:.4f} ms ({r31.n_ticks} ticks)", flush=True)
