"""Generates tests/golden/*.npz from the oracle (rcp_mode = EXACT: IEEE arithmetic only, portable across hosts).

The reference ships no fixtures for this path, so these vectors pin the *restatement* against
regressions and against the HIP path; they are not outputs of the reference itself ("parity unpinned").

Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from dvo_slam_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

CASES = {
    # name: (w, h, levels, first, last, xi scale, mu, use_init)
    "pair_160x120_l3": (160, 120, 3, 2, 0, 0.5, 0.0, False),
    "pair_320x240_l4_mu": (320, 240, 4, 3, 1, 1.0, 0.05, True),
}


def run_case(name):
    w, h, levels, first, last, s, mu, use_init = CASES[name]
    (Ir, Zr), (Ic, Zc), Tgt = synth.make_pair(w, h, xi_gt=synth.XI_GT_PAIR * s)
    K = synth.intrinsics_for(w, h)
    pr, pc = orc.Pyramid(Ir, Zr, K, levels), orc.Pyramid(Ic, Zc, K, levels)
    cfg = orc.default_config(first_level=first, last_level=last, rcp_mode=orc.RCP_EXACT, mu=mu,
                             use_initial_estimate=int(use_init), max_iterations_per_level=50)
    T0 = synth.se3_exp(synth.XI_GT_PAIR * s * 0.8) if use_init else None
    r = orc.match(cfg, pr, pc, T0)
    out = {"T": r["T"], "information": r["information"], "loglik": np.float64(r["loglik"]), "T_gt": Tgt,
           "levels": np.array([[L["id"], L["valid_pixels"], L["termination"], len(L["iterations"])] for L in r["levels"]])}
    rows = []
    for L in r["levels"]:
        for it in L["iterations"]:
            rows.append(np.concatenate([[L["id"], it["id"], it["valid_constraints"], it["tdist_loglik"], it["has_increment"]],
                                        it["precision"].ravel(), it["increment"] if it["has_increment"] else np.zeros(6)]))
    out["iterations"] = np.array(rows)
    # stage vectors of the finest level at the identity: residual checksum + a strided sample
    pe, res, valid = orc.compute_residuals(pr, pc, last, np.eye(4), orc.RCP_EXACT)
    out["res_count"] = np.int64(len(res))
    out["res_sum"] = res.astype(np.float64).sum(axis=0)
    out["res_sample"] = res[:: max(1, len(res) // 257)].copy()
    out["sel_counts"] = np.array([pr.select(l)[0].shape[0] for l in range(levels)])
    out["plane_sums"] = np.array([[np.nansum(pr.plane(l, p).astype(np.float64)) for p in range(6)] for l in range(levels)])
    return out


if __name__ == "__main__":
    for name in CASES:
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **run_case(name))
        print("wrote", name)
