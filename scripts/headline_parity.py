"""Pose parity of the headline pair (BASELINE config 2) and of 1280x960 (config 3) against both oracle modes."""
import sys

sys.path.insert(0, ".")
import numpy as np

from dvo_slam_amd import capi, synth
from oracle import oracle as orc

for (w, h, levels) in ((640, 480, 4), (1280, 960, 5)):
    (Ir, Zr), (Ic, Zc), Tgt = synth.make_pair(w, h)
    K = synth.intrinsics_for(w, h)
    g = capi.DenseTracker(capi.Config(FirstLevel=levels - 1, LastLevel=0)).match(
        capi.RgbdImagePyramid(Ir, Zr, K, levels), capi.RgbdImagePyramid(Ic, Zc, K, levels))
    pr, pc = orc.Pyramid(Ir, Zr, K, levels), orc.Pyramid(Ic, Zc, K, levels)
    line = [f"{w}x{h}: |log(Tgt^-1 T_gpu)| = {synth.pose_error(Tgt, g.Transformation):.2e}"]
    for name, mode in (("exact-reciprocal oracle", orc.RCP_EXACT), ("rcpps oracle (this host)", orc.RCP_SSE),
                       ("CLEAN oracle (no Q5 / Q6)", orc.RCP_CLEAN)):
        o = orc.match(orc.default_config(first_level=levels - 1, last_level=0, rcp_mode=mode), pr, pc)
        same = all(Lg["TerminationCriterion"] == Lo["termination"] and len(Lg["Iterations"]) == len(Lo["iterations"])
                   for Lg, Lo in zip(g.Levels, o["levels"]))
        line.append(f"vs {name}: {synth.pose_error(o['T'], g.Transformation):.2e} ({'same' if same else 'forked'} iteration path)")
    print("; ".join(line))
