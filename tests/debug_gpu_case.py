"""Print GPU and oracle iteration traces side by side for a few pairs (debug aid)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvo_slam_amd import synth, capi
from oracle import oracle as orc

def trace(name, gr, gc, orr, occ, **kw):
    cfg = capi.Config(**kw)
    trk = capi.DenseTracker(cfg)
    rg = trk.match(gr, gc)
    ro = orc.match(orc.default_config(first_level=cfg.FirstLevel, last_level=cfg.LastLevel, rcp_mode=orc.RCP_EXACT), orr, occ)
    print("==", name, "pose err", synth.pose_error(ro["T"], rg.Transformation))
    for Lg, Lo in zip(rg.Levels, ro["levels"]):
        print(" L", Lg["Id"], "gpu", capi.TERMINATION[Lg["TerminationCriterion"]], len(Lg["Iterations"]), "orc", orc.TERMINATION[Lo["termination"]], len(Lo["iterations"]))
        for k in range(max(len(Lg["Iterations"]), len(Lo["iterations"]))):
            g = Lg["Iterations"][k] if k < len(Lg["Iterations"]) else None
            o = Lo["iterations"][k] if k < len(Lo["iterations"]) else None
            gs = f"n={g['ValidConstraints']} ll={g['TDistributionLogLikelihood']:.2f} |x|={np.abs(g['EstimateIncrement']).max():.3e}" if g else "-"
            os_ = f"n={o['valid_constraints']} ll={o['tdist_loglik']:.2f} |x|={np.abs(o['increment']).max():.3e}" if o else "-"
            print(f"    it{k}: gpu {gs:55s} orc {os_}")

w, h = 640, 480
K = synth.intrinsics_for(w, h)
(Ir, Zr), (Ic, Zc), Tgt = synth.make_pair(w, h)
gr, gc = capi.RgbdImagePyramid(Ir, Zr, K, 4), capi.RgbdImagePyramid(Ic, Zc, K, 4)
orr, occ = orc.Pyramid(Ir, Zr, K, 4), orc.Pyramid(Ic, Zc, K, 4)
trace("swapped", gc, gr, occ, orr, FirstLevel=3, LastLevel=0)
T2 = synth.se3_exp([0.04, -0.02, 0.03, 0.015, -0.02, 0.01])
cur = synth.render(640, 480, T2, frame_id=5)
g2, o2 = capi.RgbdImagePyramid(cur[0], cur[1], K, 4), orc.Pyramid(cur[0], cur[1], K, 4)
trace("larger motion", gr, g2, orr, o2, FirstLevel=3, LastLevel=0)
