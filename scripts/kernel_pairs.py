"""Isolated residual pass over N DISTINCT (reference, current) pairs (nothing shared between the items of a launch) next to
the same launch over N copies of one pair.  usage: kernel_pairs.py [n_pairs=36] [reps=20]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvo_slam_amd import capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 36
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
W, H = 640, 480
K = synth.intrinsics_for(W, H)
refs = [capi.RgbdImagePyramid(*synth.render(W, H, synth.se3_exp(synth.XI_GT_PAIR * 0.05 * i), frame_id=1000 + i), K, 4) for i in range(n)]
curs = [capi.RgbdImagePyramid(*synth.render(W, H, synth.se3_exp(synth.XI_GT_PAIR * (0.5 + 0.02 * i)), frame_id=1 + 2 * i), K, 4) for i in range(n)]
trk = capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=0))
T = synth.se3_exp(synth.XI_GT_PAIR * 0.6)
tag = f"lib={os.path.basename(os.environ.get('DVO_AMD_LIB', 'default'))} accum={os.environ.get('DVO_AMD_ACCUM', 'mfma16')} occ={os.environ.get('DVO_AMD_OCC', '4')}"
for level in (0, 1, 2, 3):
    ms, ab, nl = trk.bench_residual_pass_pairs(refs, curs, level, T, 0, reps)
    ms1, ab1, nl1 = trk.bench_residual_pass(refs[0], curs[0], level, T, n, 0, reps)
    print(f"{tag} level {level}: {n} distinct pairs {ms*1e3:7.1f} us {ab/ms/1e6:7.1f} GB/s ({ab/ms/1e6/8000:.3f}) | one pair x{n} {ms1*1e3:7.1f} us "
          f"{ab1/ms1/1e6:7.1f} GB/s ({ab1/ms1/1e6/8000:.3f})", flush=True)
