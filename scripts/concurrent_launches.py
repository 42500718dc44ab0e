"""Do k_tick launches of different streams overlap usefully?  N host threads, one tracker (stream) each, every thread launches
the same residual pass (level L, n items) `reps` times back to back; per-launch duration (dispatch stamps) and the aggregate
rate against one thread's.  usage: concurrent_launches.py"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvo_slam_amd import capi, synth

K = synth.intrinsics_for(640, 480)
(Ir, Zr), (Ic, Zc), Tgt = synth.make_pair(640, 480)
pr, pc = capi.RgbdImagePyramid(Ir, Zr, K, 4), capi.RgbdImagePyramid(Ic, Zc, K, 4)
REPS = 300
for level, n_items in ((0, 4), (1, 8), (1, 16), (2, 16), (0, 12)):
    base = None
    for n_threads in (1, 2, 3, 4, 6, 8):
        trackers = [capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=0)) for _ in range(n_threads)]
        for t in trackers:
            t.bench_residual_pass(pr, pc, level, Tgt, n_items, 0, reps=5)
        out = [None] * n_threads

        def work(i):
            out[i] = trackers[i].bench_residual_pass(pr, pc, level, Tgt, n_items, 0, reps=REPS)
        th = [threading.Thread(target=work, args=(i,)) for i in range(n_threads)]
        t0 = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        wall = time.perf_counter() - t0
        per_launch_us = sum(o[0] for o in out) / n_threads * 1e3
        mb = out[0][1] / 1e6
        agg = n_threads * REPS * mb / wall / 1e6  # TB/s
        base = base or agg
        print(f"level {level} items {n_items:2d} ({mb:6.1f} MB/launch) threads {n_threads}: kernel {per_launch_us:7.2f} us each, "
              f"wall per launch-slot {wall / REPS * 1e6:7.2f} us, aggregate {agg:5.2f} TB/s ({agg / base:4.2f}x one thread)", flush=True)
