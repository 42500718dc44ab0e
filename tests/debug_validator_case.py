"""GPU debugging aid (not a test): one proposal of a validator scenario, single match() on both sides with full statistics and
the fork adjudication, next to the batched validator's result for the same proposal.
usage: python tests/debug_validator_case.py [origin] [n_candidates] [first,last]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fork_criterion  # noqa: E402
import validator_scenario as S  # noqa: E402
from dvo_slam_amd import capi, constraints as Cn, synth  # noqa: E402
from oracle import oracle as orc, validator as V  # noqa: E402

origin = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n_cand = int(sys.argv[2]) if len(sys.argv) > 2 else 6
levels = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else None
decoys = n_cand != 32
key, cands = synth.loop_closure_scenario(640, 480, n_cand, decoys=decoys) if not decoys else synth.loop_closure_scenario(640, 480, n_cand)
K = synth.intrinsics_for(640, 480)
ents = [key] + cands
opyr = {e["id"]: orc.Pyramid(e["frame"][0], e["frame"][1], K, 4) for e in ents}
gpyr = {e["id"]: capi.RgbdImagePyramid(e["frame"][0], e["frame"][1], K, 4) for e in ents}
idx = origin if origin >= 0 else -origin - 1
cand = cands[idx // 2]
init = np.eye(4) if idx % 2 == 0 else np.linalg.inv(cand["pose"]) @ key["pose"]
ref, cur = key["id"], cand["id"]
if origin < 0:
    ref, cur, init = cur, ref, np.linalg.inv(init)
ov = V.create_constraint_proposal_validator(min_constraint_ratio=0.0, ratio_coarse=-1e300, ratio_fine=-1e300)
ocfg = ov.stages[0].TrackingConfig
if levels:
    ocfg.first_level, ocfg.last_level = levels
print("proposal", ref, "->", cur, "stage config", ocfg.first_level, ocfg.last_level, ocfg.use_initial_estimate, ocfg.max_iterations_per_level)
ro = orc.match(ocfg, opyr[ref], opyr[cur], init)
gcfg = capi.Config(FirstLevel=ocfg.first_level, LastLevel=ocfg.last_level, UseInitialEstimate=bool(ocfg.use_initial_estimate),
                   MaxIterationsPerLevel=ocfg.max_iterations_per_level, Precision=ocfg.precision, Mu=ocfg.mu)
trk = capi.DenseTracker(gcfg)
rg = trk.match(gpyr[ref], gpyr[cur], init)
err = synth.pose_error(ro["T"], rg.Transformation)
print("single match: pose error vs oracle", err)
# the same pair inside batches of different sizes (other wave-segment lengths, other summation order)
for n in (9, 40, 130):
    out = trk.match_batch([gpyr[ref]] * n, [gpyr[cur]] * n, T_inits=[init] * n, in_flight=72)
    errs = sorted({round(synth.pose_error(ro["T"], o.Transformation), 9) for o in out})
    print(f"batch of {n}: pose errors vs oracle {errs}; paths {sorted({tuple((L['TerminationCriterion'], len(L['Iterations'])) for L in o.Levels) for o in out})}")
for Lg, Lo in zip(rg.Levels, ro["levels"]):
    print("level", Lg["Id"], "GPU", Lg["TerminationCriterion"], len(Lg["Iterations"]), "oracle", Lo["termination"], len(Lo["iterations"]))
    for k in range(max(len(Lg["Iterations"]), len(Lo["iterations"]))):
        g = Lg["Iterations"][k] if k < len(Lg["Iterations"]) else None
        o = Lo["iterations"][k] if k < len(Lo["iterations"]) else None
        print(f"  {k}: GPU", None if g is None else (g["ValidConstraints"], round(g["TDistributionLogLikelihood"], 3), float(np.abs(g["EstimateIncrement"]).max())),
              "oracle", None if o is None else (o["valid_constraints"], round(o["tdist_loglik"], 3), float(np.abs(o["increment"]).max())),
              "pose gap", None if g is None or o is None else synth.pose_error(g["estimate"], o["estimate"]))
for name, mode in (("fp64", orc.SUM_FP64), ("blocked", orc.SUM_BLOCKED)):
    kw = {f: getattr(ocfg, f) for f, _ in ocfg._fields_}
    kw["sum_mode"] = mode
    r = orc.match(orc.default_config(**kw), opyr[ref], opyr[cur], init)
    print("oracle", name, "self distance", synth.pose_error(ro["T"], r["T"]), [(L["termination"], len(L["iterations"])) for L in r["levels"]])
# and under tiny perturbations of the initial transform (the chaos test's probe)
ds = []
for k in range(12):
    xi = np.zeros(6)
    xi[k % 6] = 1e-9 * (1 + k)
    ds.append(synth.pose_error(ro["T"], orc.match(ocfg, opyr[ref], opyr[cur], synth.se3_exp(xi) @ init)["T"]))
print("oracle under 1e-9 perturbations of the initial transform:", ["%.1e" % d for d in ds])
try:
    for line in fork_criterion.adjudicate(orc, synth, ocfg, opyr[ref], opyr[cur], init, rg, ro, err, 1e-5)[0]:
        print(line)
except AssertionError as exc:
    print("ADJUDICATION FAILED:", exc)
