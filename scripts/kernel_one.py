"""Run the isolated residual-pass kernel for one configuration (for rocprofv3 --pmc)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvo_slam_amd import synth, capi
level, n_items, rounds, reps = [int(a) for a in sys.argv[1:5]]
w, h = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (640, 480)
levels = 5 if w >= 1280 else 4
K = synth.intrinsics_for(w, h)
(Ir, Zr), (Ic, Zc), Tgt = synth.make_pair(w, h)
pr, pc = capi.RgbdImagePyramid(Ir, Zr, K, levels), capi.RgbdImagePyramid(Ic, Zc, K, levels)
trk = capi.DenseTracker(capi.Config(FirstLevel=levels - 1, LastLevel=0))
ms, ab, nl = trk.bench_residual_pass(pr, pc, level, Tgt, n_items, rounds, reps=reps)
print(f"level {level} items {n_items} rounds {rounds}: {ms*1e3:.1f} us {ab/ms/1e6:.1f} GB/s launches {nl}")
