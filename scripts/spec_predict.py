"""Data for a predictor of the likelihood test's outcome: for every iteration k >= 1 of the bench's pairs, the norm of the increment
that led to it, the likelihood gain of the iteration before, and whether iteration k was accepted (its likelihood did not
decrease) or ended the level."""
import sys
sys.path.insert(0, ".")
import numpy as np
from dvo_slam_amd import capi, synth
W, H = 640, 480
K = synth.intrinsics_for(W, H)
trk = capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=0))
rows = []
def cur_pose(i):  # bench.py's frames
    return synth.se3_exp(synth.XI_GT_PAIR * (0.5 + 0.9 * ((i * 7) % 13) / 13.0) * (1 if i % 2 == 0 else -1)
                         + synth.XI_GT_PAIR[::-1] * 0.03 * ((i * 5) % 11 - 5))
refs = [capi.RgbdImagePyramid(*synth.render(W, H, None if r == 0 else synth.se3_exp(synth.XI_GT_PAIR * 0.05 * r), frame_id=(0 if r == 0 else 1000 + r)), K, 4) for r in range(4)]
curs = [capi.RgbdImagePyramid(*synth.render(W, H, cur_pose(i), frame_id=1 + 2 * i), K, 4) for i in range(24)]
for r, ref in enumerate(refs):
    for i, cur in enumerate(curs):
        res = trk.match(ref, cur)
        for L in res.Levels:
            its = L["Iterations"]
            for k in range(1, len(its)):
                inc = np.linalg.norm(its[k - 1]["EstimateIncrement"]) if its[k - 1]["has_increment"] else np.nan
                gain_prev = (its[k - 1]["TDistributionLogLikelihood"] - its[k - 2]["TDistributionLogLikelihood"]) if k >= 2 else np.nan
                accepted = its[k]["TDistributionLogLikelihood"] >= its[k - 1]["TDistributionLogLikelihood"]
                last = k == len(its) - 1
                rows.append((L["Id"], k, inc, gain_prev, abs(its[k - 1]["TDistributionLogLikelihood"]), accepted, last, L["TerminationCriterion"]))
rows = np.array(rows, dtype=float)
# the statistics carry the NEGATIVE log-likelihood: an iteration is accepted when the stored value did not increase
rows[:, 5] = 1 - rows[:, 5]
px = {3: 4800, 2: 19200, 1: 76800, 0: 307200}
for levels in ((0, 1, 2, 3), (0, 1), (0,)):
    sel = rows[np.isin(rows[:, 0], levels)]
    acc, rej = sel[sel[:, 5] == 1], sel[sel[:, 5] == 0]
    print(f"levels {levels}: iterations k>=1: {len(sel)}, accepted {len(acc)}, rejected {len(rej)} (each wastes one speculative residual pass)")
    for thr in (3e-6, 1e-5, 3e-5, 1e-4, 3e-4, 1e-3):
        wa = np.array([px[int(l)] for l in rej[:, 0]])
        saved = wa[rej[:, 2] < thr].sum() / max(wa.sum(), 1)
        pass
    # second feature: the likelihood gain of the iteration before, relative to the likelihood
    for g in (1e-6, 1e-5, 1e-4):
        wa = np.array([px[int(l)] for l in rej[:, 0]])
        fr, fa = -rej[:, 3] / rej[:, 4] < g, -acc[:, 3] / acc[:, 4] < g
        print(f"   'previous relative gain < {g:g} -> no speculation': avoids {np.nanmean(fr):.2f} of the wasted passes ({wa[fr].sum() / max(wa.sum(), 1):.2f} of their pixels), extra tick at {np.nanmean(fa):.3f} of the accepted")
    for thr in (3e-6, 1e-5, 3e-5, 1e-4, 3e-4, 1e-3):
        wa = np.array([px[int(l)] for l in rej[:, 0]])
        saved = wa[rej[:, 2] < thr].sum() / max(wa.sum(), 1)
        print(f"   'increment norm < {thr:g} -> no speculation': avoids {np.mean(rej[:, 2] < thr):.2f} of the wasted passes ({saved:.2f} of their pixels), "
              f"costs an extra tick at {np.mean(acc[:, 2] < thr):.3f} of the accepted iterations ({int(np.sum(acc[:, 2] < thr))} of {len(acc)})")
