import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvo_slam_amd import synth, capi
K = synth.intrinsics_for(640, 480)
(Ir, Zr), (Ic, Zc), Tgt = synth.make_pair(640, 480)
pr, pc = capi.RgbdImagePyramid(Ir, Zr, K, 4), capi.RgbdImagePyramid(Ic, Zc, K, 4)
trk = capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=0))
t0 = time.time()
while time.time() - t0 < float(sys.argv[1]):
    ms, ab, nl = trk.bench_residual_pass(pr, pc, 0, Tgt, 36, 0, reps=200)
print(f"level-0 pass alone, 36 pairs per launch, back to back for {sys.argv[1]} s: last {ms*1e3:.1f} us per launch, {ab/ms/1e6:.0f} GB/s")
