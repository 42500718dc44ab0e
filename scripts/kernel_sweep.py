"""Sweep the isolated residual-pass kernel over items / rounds / levels; prints GB/s of algorithmic bytes."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvo_slam_amd import synth, capi
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (640, 480)
levels = 5 if w >= 1280 else 4
K = synth.intrinsics_for(w, h)
(Ir, Zr), (Ic, Zc), Tgt = synth.make_pair(w, h)
pr, pc = capi.RgbdImagePyramid(Ir, Zr, K, levels), capi.RgbdImagePyramid(Ic, Zc, K, levels)
trk = capi.DenseTracker(capi.Config(FirstLevel=levels - 1, LastLevel=0))
print("mode", os.environ.get("DVO_AMD_ACCUM", "mfma"))
for level in (0, 1, 3):
    for n_items in (1, 8, 30, 64, 128):
        for rounds in (1, 2, 4, 8, 16):
            if n_items * (w * h >> (2 * level)) // (1024 * rounds) < 1 and rounds > 1:
                continue
            ms, ab, nl = trk.bench_residual_pass(pr, pc, level, Tgt, n_items, rounds, reps=10)
            print(f"level {level} items {n_items:4d} rounds {rounds:2d} launches {nl}: {ms*1e3:9.1f} us  {ab/ms/1e6:8.1f} GB/s  ({ab/ms/1e6/8000*100:5.1f}% of 8 TB/s)", flush=True)
