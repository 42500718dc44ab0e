"""How much residual-pass work the speculative scheme discards on the bench workload (per pair: iterations per level,
termination per level, residual passes run, ticks)."""
import sys

sys.path.insert(0, ".")
from dvo_slam_amd import capi, synth

W, H = 640, 480
K = synth.intrinsics_for(W, H)
ref = capi.RgbdImagePyramid(*synth.render(W, H, None, frame_id=0), K, 4)
trk = capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=0))
tot_it = tot_pass = 0
px = {3: 4800, 2: 19200, 1: 76800, 0: 307200}
useful = wasted = 0.0
for i in range(8):
    xi = synth.XI_GT_PAIR * (0.6 + 0.1 * i) * (1 if i % 2 == 0 else -1)
    cur = capi.RgbdImagePyramid(*synth.render(W, H, synth.se3_exp(xi), frame_id=1 + 2 * i), K, 4)
    r = trk.match(ref, cur)
    its = [len(L["Iterations"]) for L in r.Levels]
    term = [L["TerminationCriterion"] for L in r.Levels]
    for L in r.Levels:
        n = len(L["Iterations"])
        # a level that ends with LogLikelihoodDecreased (2) ran one residual pass whose result is discarded
        useful += n * px[L["Id"]]
        if L["TerminationCriterion"] == 2:
            wasted += px[L["Id"]]
    print(f"pair {i}: iterations {its} termination {term} residual passes {r.n_residual_passes} ticks {r.n_ticks}")
    tot_it += sum(its)
    tot_pass += r.n_residual_passes
print(f"iterations {tot_it}, residual passes {tot_pass}; pixel-weighted: speculative passes = {wasted / (useful + wasted):.3f} of all residual-pass work")
