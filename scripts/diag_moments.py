"""Diagnostic: where do the GPU's normal equations differ from the oracle's at a near-converged iteration?"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvo_slam_amd import capi, synth
from oracle import oracle as orc
w, h = 640, 480
(Ir, Zr), (Ic, Zc), Tgt = synth.make_pair(w, h)
K = synth.intrinsics_for(w, h)
pr, pc = orc.Pyramid(Ir, Zr, K, 4), orc.Pyramid(Ic, Zc, K, 4)
gr, gc = capi.RgbdImagePyramid(Ir, Zr, K, 4), capi.RgbdImagePyramid(Ic, Zc, K, 4)
trk = capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=0))
ro = orc.match(orc.default_config(first_level=3, last_level=0, rcp_mode=orc.RCP_EXACT), pr, pc)
for L in ro["levels"]:
    prec = None
    for k, it in enumerate(L["iterations"]):
        if not it["has_increment"]:
            prec = it["precision"]; continue
        g_own = trk.iteration_probe(gr, gc, L["id"], it["estimate"], prec)
        g_orc = trk.iteration_probe(gr, gc, L["id"], it["estimate"], prec, it["precision"])
        b, A = it["rhs"], it["information"]
        pe, res, valid = orc.compute_residuals(pr, pc, L["id"], it["estimate"], orc.RCP_EXACT)
        n = len(res)
        scale = np.sqrt(np.diag(A) * 2 * n)
        dP = np.abs(g_own["precision"] - it["precision"]).max() / np.abs(it["precision"]).max()
        print(f"L{L['id']} k{k} n{n}: dP/P {dP:.1e} | b own-P: max|db|/max|b| {np.abs(g_own['b']-b).max()/np.abs(b).max():.1e} "
              f"| b oracle-P: {np.abs(g_orc['b']-b).max()/np.abs(b).max():.1e}  /CS-scale {(np.abs(g_orc['b']-b)/scale).max():.1e} "
              f"| A oracle-P rel {np.abs(g_orc['A']-A).max()/np.abs(A).max():.1e} | ll own {abs(-g_own['ll']-it['tdist_loglik'])/abs(it['tdist_loglik']):.1e} "
              f"orcP {abs(-g_orc['ll']-it['tdist_loglik'])/abs(it['tdist_loglik']):.1e}")
        prec = it["precision"]
