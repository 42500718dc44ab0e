"""The residual pass alone as a function of the launch size: one level, n copies of one pair per launch (n = 1 .. 36), the
level's own wave-segment geometry.  Fits  us = a + b * MB  per level: a = what a launch costs before and after it streams
(ramp, the first dependent load chains of its blocks, the tail), 1 / b = the marginal rate.  usage: launch_size_curve.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvo_slam_amd import capi, synth

K = synth.intrinsics_for(640, 480)
(Ir, Zr), (Ic, Zc), Tgt = synth.make_pair(640, 480)
pr, pc = capi.RgbdImagePyramid(Ir, Zr, K, 4), capi.RgbdImagePyramid(Ic, Zc, K, 4)
trk = capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=0))
for level in (0, 1, 2, 3):
    pts = []
    for n in (1, 2, 4, 8, 12, 16, 24, 36):
        ms, ab, nl = trk.bench_residual_pass(pr, pc, level, Tgt, n, 0, reps=30)
        pts.append((ab / 1e6, ms * 1e3))
        print(f"level {level} items {n:3d}: {ms * 1e3:8.2f} us  {ab / 1e6:8.2f} MB  {ab / ms / 1e9:6.2f} TB/s", flush=True)
    mb, us = np.array(pts).T
    b, a = np.polyfit(mb, us, 1)
    print(f"level {level}: us = {a:.2f} + {b:.4f} * MB  (marginal {1 / b / 1e3:.2f} TB/s... {1e-6 / (b * 1e-6) / 1e6:.2f} MB/us)", flush=True)
