"""What makes a fork legitimate.

A free-running match() on the GPU and the oracle sometimes part ways: one side runs an iteration more on some level, or ends
it for another reason.  Rounds 1-2 bounded that with a count of forked configurations; a count cannot tell a coin flip from
a bug.  This module adjudicates every fork on its own evidence (dense_tracking.cpp:304-322 and :357-363 are the two tests
that can flip):

1. SELF-DISTANCE.  The oracle is run again with the SAME fp32 terms of its scale / normal-equation sums added up in a different
   order (orc_config.sum_mode: fp64 accumulator, blocked fp32 partial sums).  How far those runs land from the reference-order
   run is how far the reference algorithm moves under re-association alone; the GPU may be that far from the oracle (times
   SELF_DISTANCE_SLACK), and no further, whatever path it took.

2. THE FLIPPED DECISION.  At the first iteration where the two sides decided differently:
   a. "summation noise": the oracle's own margin (|ll_k - ll_{k-1}| for an accept / reject flip, | |x|_inf - Precision | for a
      stop / continue flip) is inside the band by which its sequential fp32 sums miss the exact sums -- measured with the float64
      restatement (tests/stage_f64.py) at the oracle's own poses.  The oracle's decision is then a coin flip of its own rounding.
   b. "pose drift": otherwise the two sides must have reached that iteration with different inputs (poses that differ by ~1e-7,
      which the reference algorithm amplifies through quirks Q5 / Q6).  Then the reference arithmetic (orc.iteration) is evaluated
      at the GPU's OWN poses of iterations k-1 and k: it must see the GPU's constraint counts, reproduce the GPU's likelihoods
      within the reference's summation-noise band there, and take the GPU's decision (or be undecided within that band).
   A flip that is neither fails the test.

3. SELF-CONSISTENCY.  Each side's recorded iteration counts and termination criteria must be what the reference's control flow
   produces from that side's own recorded numbers (replay_level).

One artefact of the reference is worth a note when it occurs (overflow_note): computeCompleteDataLogLikelihood multiplies 50 terms
(1 + 0.2 r^T P r) in a double before it takes a log (dense_tracking_impl.cpp:413-419).  When 50 consecutive residuals all have a
Mahalanobis distance above ~7e6 that product overflows, the reference's likelihood is -inf and it rejects the iteration.  This
needs precisions of 1e9 and more, which only noise-free synthetic depth produces (2 of the 64 alignments of BASELINE config 5's
scenario; never on sensor data, where the depth precision is ~1e4).  The GPU path reproduces it: its likelihood pass reports the
largest Mahalanobis distance it saw and, when a group of fifty could have overflowed, k_ll_overflow redoes the reference's own
multiplications (csrc/dvo_tracker.cpp: ll_overflowed).  The oracle's `ll_guard` mode -- the same sum without the overflow --
shows how far the artefact moves the answer (tests/test_gpu_parity.py::test_overflowing_likelihood_is_reproduced).  The one path
that does NOT emulate it is a pair tile-sharded over several GPUs (each rank holds only its band's residuals).
"""
import numpy as np

from stage_f64 import f64_iteration

SELF_DISTANCE_SLACK = 2.0
TERM_NAMES = {0: "IterationsExceeded", 1: "IncrementTooSmall", 2: "LogLikelihoodDecreased", 3: "TooFewConstraints", -1: "Unset"}


def ulp32(x):
    return float(np.spacing(np.float32(abs(x))))


def gpu_levels(rg):
    return [dict(id=L["Id"], termination=L["TerminationCriterion"],
                 iters=[dict(V=it["ValidConstraints"], nll=it["TDistributionLogLikelihood"], has_inc=it["has_increment"],
                             inc=it["EstimateIncrement"], P=np.asarray(it["TDistributionPrecision"], np.float32),
                             T=it["estimate"], initial=it["initial"]) for it in L["Iterations"]]) for L in rg.Levels]


def oracle_levels(ro):
    return [dict(id=L["id"], termination=L["termination"],
                 iters=[dict(V=it["valid_constraints"], nll=it["tdist_loglik"], has_inc=bool(it["has_increment"]),
                             inc=it["increment"], P=np.asarray(it["precision"], np.float32), T=it["estimate"],
                             rhs=it["rhs"]) for it in L["iterations"]]) for L in ro["levels"]]


def replay_level(iters, precision, max_iter):
    """dense_tracking.cpp:273-363 re-run on the recorded numbers of one level: (iterations, set of terminations the rule allows).
    None if the list ends although the rule would go on."""
    last, x, n, term, accepted = np.inf, None, None, None, 0
    for k, it in enumerate(iters):
        if it["V"] < 6:  # :276-284
            n, term = k + 1, 3
            break
        if not (it["nll"] < last):  # :312-322 (NaN compares false: rejected)
            n, term = k + 1, 2
            break
        last = it["nll"]
        x = it["inc"]
        accepted = k + 1
        if not (np.abs(x).max() > precision and accepted < max_iter):  # the while condition, :357
            n = k + 1
            break
    if n is None:
        return None
    allowed = {term} if term is not None else set()
    # the overrides behind the loop are evaluated after a break as well (:359-363)
    if x is None:
        allowed.add(1)  # x is still the level's start value log(inc) (Q1): not in the statistics, may be below Precision
    elif np.abs(x).max() <= precision:
        allowed = {1}
    if accepted >= max_iter:
        allowed = {0}
    return n, allowed


def check_self_consistency(levels, precision, max_iter, who):
    for L in levels:
        r = replay_level(L["iters"], precision, max_iter)
        assert r is not None, (who, "level", L["id"], "ends although its own numbers say continue")
        n, allowed = r
        assert n == len(L["iters"]) and L["termination"] in allowed, \
            (who, "level", L["id"], "recorded", len(L["iters"]), TERM_NAMES[L["termination"]], "its own numbers give", n,
             [TERM_NAMES[t] for t in allowed])


def self_distance(orc, synth, ocfg, o_ref, o_cur, T_init, ro, thorough=False):
    """How far the oracle lands from itself under perturbations SMALLER than what separates the GPU's arithmetic from the
    reference's: the same fp32 terms of its sums added up in another order (an fp64 accumulator; blocked fp32 partial sums) and,
    with thorough=True, three block sizes plus the probe of tests/test_oracle.py::test_reference_algorithm_is_chaotic -- the
    initial transform moved by 1e-9 along each axis (the reference's outcomes are heavy-tailed: two samples under-estimate the
    spread).  Returns {variant: (distance, iteration path)}."""
    base = {f: getattr(ocfg, f) for f, _ in ocfg._fields_}
    variants = [("fp64 accumulator", dict(sum_mode=orc.SUM_FP64), None), ("blocked fp32", dict(sum_mode=orc.SUM_BLOCKED), None)]
    if thorough:
        variants += [("blocked fp32 / 32", dict(sum_mode=orc.SUM_BLOCKED_32), None),
                     ("blocked fp32 / 2048", dict(sum_mode=orc.SUM_BLOCKED_2048), None)]
        T0 = np.eye(4) if (T_init is None or not ocfg.use_initial_estimate) else np.asarray(T_init, np.float64)
        for k in range(6):
            xi = np.zeros(6)
            xi[k] = 1e-9 * (1 + k)
            variants.append((f"initial transform moved 1e-9 along axis {k}", dict(use_initial_estimate=1), orc.se3_exp(xi) @ T0))
    out = {}
    for name, kw, T in variants:
        r = orc.match(orc.default_config(**dict(base, **kw)), o_ref, o_cur, T_init if T is None else T)
        out[name] = (synth.pose_error(ro["T"], r["T"]),
                     [(TERM_NAMES[L["termination"]], len(L["iterations"])) for L in r["levels"]])
    return out


def has_overflowed_likelihood(ro):
    return any(not np.isfinite(it["tdist_loglik"]) for L in ro["levels"] for it in L["iterations"] if it["valid_constraints"] >= 6)


def overflow_note(ro):
    """A note when one of the oracle's likelihoods overflowed to -inf (module docstring); the comparison itself is unchanged:
    the GPU path reproduces the artefact."""
    if not has_overflowed_likelihood(ro):
        return None
    where = [(L["id"], k) for L in ro["levels"] for k, it in enumerate(L["iterations"]) if not np.isfinite(it["tdist_loglik"])]
    return f"reference overflow artefact at (level, iteration) {where}: the reference's 50-term likelihood product overflowed to " \
           f"inf there and the iteration was rejected; the GPU result is held to the same rule as everywhere (it reproduces this)"


def without_overflow(orc, ocfg, o_ref, o_cur, T_init):
    """the same configuration with the oracle's ll_guard mode: the same likelihood sum without the overflow"""
    kw = {f: getattr(ocfg, f) for f, _ in ocfg._fields_}
    kw["ll_guard"] = 1
    return orc.match(orc.default_config(**kw), o_ref, o_cur, T_init)


# No forked path may be further from the oracle than this, whatever the oracle's own spread.  3e-4 on noise-free input (the
# largest self-distance seen there is 1.7e-4).  On sensor-noise input one more or one fewer accepted step at a converged level
# moves the estimate by up to ~1e-3 -- the oracle does that to itself under a re-associated sum (7.5e-4 on a loop-closure pair,
# the GPU landing on the same outcome to three digits), and the estimator's own accuracy against ground truth is of that
# size there: callers in that regime pass SENSOR_REGIME_CEILING.
DIVERGED_PATH_CEILING = 3e-4
SENSOR_REGIME_CEILING = 3e-3


def pose_bar(orc, synth, ocfg, o_ref, o_cur, T_init, ro, err, pose_tol=1e-5, ceiling=None):
    """The pose tolerance for callers that hold no per-iteration statistics of the GPU side (the batched validator, the
    front-end step): pose_tol when the GPU is within it; otherwise the GPU may be as far from the oracle as the oracle lands
    from itself under re-associated sums (times SELF_DISTANCE_SLACK), which costs two to twelve more oracle alignments.
    Returns (bar, note)."""
    if err <= pose_tol:
        return pose_tol, None
    sd = self_distance(orc, synth, ocfg, o_ref, o_cur, T_init, ro)
    if err > SELF_DISTANCE_SLACK * max(d for d, _ in sd.values()):
        sd = self_distance(orc, synth, ocfg, o_ref, o_cur, T_init, ro, thorough=True)
    d_self = max(d for d, _ in sd.values())
    worst = max(sd, key=lambda k: sd[k][0])
    # an absolute ceiling on top: on a chaotic pair the oracle's own spread can grow to the size of a whole
    # Gauss-Newton step, and a bar that grows with it would let a real regression through
    return min(DIVERGED_PATH_CEILING if ceiling is None else ceiling, max(pose_tol, SELF_DISTANCE_SLACK * d_self)), \
        f"pose error {err:.2e}; the oracle lands up to {d_self:.2e} from itself ({worst}; {len(sd)} perturbations below the " \
        f"GPU's arithmetic differences tried)"


def _noise_band(orc, o_ref, o_cur, level, T, prec_in, ll_ref, x_ref=None, mu=0.0, prior=None, sel=(0.0, 0.0)):
    """|reference arithmetic - exact sums| for the likelihood (and, if x_ref is given, the increment) of one iteration at pose T:
    the float64 restatement recomputes the scale from exact sums, inverts it, and evaluates likelihood / normal equations under
    it.  mu, prior = Mu and Mu * log(initial): the prior terms of A and b (dense_tracking.cpp:345-346)."""
    n, cov64, *_ = f64_iteration(orc, o_ref, o_cur, level, T, prec_in, np.eye(2), *sel)
    P64 = np.linalg.inv(cov64)
    n, _, A64, b64, _, ll64 = f64_iteration(orc, o_ref, o_cur, level, T, prec_in, P64, *sel)
    band_ll = abs(ll_ref - ll64)
    band_x = None
    if x_ref is not None:
        pr = np.zeros(6) if prior is None else prior
        band_x = float(np.abs(np.linalg.solve(A64 + mu * np.eye(6), b64 + pr) - x_ref).max())
    return n, band_ll, band_x


def adjudicate(orc, synth, ocfg, o_ref, o_cur, T_init, rg, ro, err, pose_tol):
    """Returns a list of report lines; raises AssertionError if the fork is not legitimate."""
    G, O = gpu_levels(rg), oracle_levels(ro)
    precision, max_iter, mu = ocfg.precision, ocfg.max_iterations_per_level, ocfg.mu
    sel = (float(ocfg.intensity_derivative_threshold), float(ocfg.depth_derivative_threshold))  # the point selection in force
    check_self_consistency(G, precision, max_iter, "GPU")
    check_self_consistency(O, precision, max_iter, "oracle")
    report = []
    # ---- 1. self-distance of the reference algorithm under re-association
    sd = self_distance(orc, synth, ocfg, o_ref, o_cur, T_init, ro)
    if err > max(pose_tol, SELF_DISTANCE_SLACK * max(d for d, _ in sd.values())):
        sd = self_distance(orc, synth, ocfg, o_ref, o_cur, T_init, ro, thorough=True)
    d_self = max(d for d, _ in sd.values())
    report.append(f"pose error vs oracle {err:.2e}; the oracle under re-associated sums lands "
                  + ", ".join(f"{d:.2e} ({name}: {path})" for name, (d, path) in sd.items()) + " from itself")
    assert err <= max(pose_tol, SELF_DISTANCE_SLACK * d_self), \
        ("forked AND further from the oracle than the oracle is from itself under re-association", err, sd)
    # ---- 2. the first decision that differs
    li = next(i for i, (a, b) in enumerate(zip(G, O)) if len(a["iters"]) != len(b["iters"]) or a["termination"] != b["termination"])
    Lg, Lo = G[li], O[li]
    level = Lg["id"]
    k = min(len(Lg["iters"]), len(Lo["iters"])) - 1
    ig, io = Lg["iters"][k], Lo["iters"][k]
    cont_g, cont_o = len(Lg["iters"]) > k + 1, len(Lo["iters"]) > k + 1
    where = f"level {level} iteration {k}: GPU {len(Lg['iters'])} iterations / {TERM_NAMES[Lg['termination']]}, " \
            f"oracle {len(Lo['iters'])} / {TERM_NAMES[Lo['termination']]}"
    if ig["has_inc"] != io["has_inc"]:
        # -- accept / reject flipped (dense_tracking.cpp:312)
        assert k >= 1, (where, "the first iteration of a level is always accepted")
        bands = []
        for j in (k - 1, k):
            pin = None if j == 0 else Lo["iters"][j - 1]["P"]
            n, b_ll, _ = _noise_band(orc, o_ref, o_cur, level, Lo["iters"][j]["T"], pin, -Lo["iters"][j]["nll"], sel=sel)
            assert n == Lo["iters"][j]["V"]
            bands.append(b_ll)
        margin = abs(Lo["iters"][k]["nll"] - Lo["iters"][k - 1]["nll"])
        noise = sum(bands) + 2 * ulp32(Lo["iters"][k]["nll"])
        if margin <= noise:
            report.append(f"{where}: accept / reject flipped by SUMMATION NOISE -- the oracle's own margin |ll_k - ll_k-1| = "
                          f"{margin:.3g} is inside the {noise:.3g} by which its sequential fp32 sums miss the exact sums there")
            return report
        # pose drift: the reference arithmetic at the GPU's own poses
        ll2, band2 = [], []
        for j in (k - 1, k):
            pin = None if j == 0 else Lg["iters"][j - 1]["P"]
            o2 = orc.iteration(o_ref, o_cur, level, Lg["iters"][j]["T"], pin, orc.RCP_EXACT, *sel)
            assert o2["n"] == Lg["iters"][j]["V"], (where, "at the GPU's pose of iteration", j, "the reference arithmetic sees",
                                                    o2["n"], "constraints, the GPU", Lg["iters"][j]["V"])
            n, b_ll, _ = _noise_band(orc, o_ref, o_cur, level, Lg["iters"][j]["T"], pin, o2["ll"], sel=sel)
            gap = abs(-Lg["iters"][j]["nll"] - o2["ll"])
            assert gap <= 2 * b_ll + 4 * ulp32(o2["ll"]), \
                (where, "iteration", j, "GPU likelihood", -Lg["iters"][j]["nll"], "reference arithmetic at the same pose", o2["ll"],
                 "its own summation-noise band", b_ll)
            ll2.append(o2["ll"])
            band2.append(b_ll)
        m2 = ll2[1] - ll2[0]  # accepted iff -ll_k < -ll_k-1
        assert (m2 > 0) == ig["has_inc"] or abs(m2) <= sum(band2) + 2 * ulp32(ll2[1]), \
            (where, "at the GPU's own poses the reference arithmetic decides", "accept" if m2 > 0 else "reject", "by", m2,
             "(band", sum(band2), ") but the GPU", "accepted" if ig["has_inc"] else "rejected")
        report.append(f"{where}: accept / reject flipped by POSE DRIFT -- oracle margin {margin:.3g} (band {noise:.3g}), but the "
                      f"two sides reached the iteration at different poses (V {ig['V']} vs {io['V']}); at the GPU's own poses the "
                      f"reference arithmetic sees the GPU's constraint counts and likelihoods and decides like the GPU "
                      f"(ll_k - ll_k-1 = {m2:.3g}, band {sum(band2):.3g})")
        return report
    if ig["has_inc"] and io["has_inc"] and cont_g != cont_o:
        # -- stop / continue flipped (dense_tracking.cpp:357: |x|_inf > Precision)
        pin = None if k == 0 else Lo["iters"][k - 1]["P"]
        # the prior term Mu * log(initial) of the oracle's right-hand side: what its recorded b_d holds beyond the data term
        prior_o = io["rhs"] - np.asarray(orc.iteration(o_ref, o_cur, level, io["T"], pin, orc.RCP_EXACT, *sel)["b"], np.float64) if mu else None
        _, _, band_x = _noise_band(orc, o_ref, o_cur, level, io["T"], pin, -io["nll"], x_ref=io["inc"], mu=mu, prior=prior_o, sel=sel)
        margin = abs(np.abs(io["inc"]).max() - precision)
        if margin <= band_x:
            report.append(f"{where}: stop / continue flipped by SUMMATION NOISE -- | |x|_inf - Precision | = {margin:.3g} on the "
                          f"oracle, its increment is {band_x:.3g} from the one exact sums give")
            return report
        ping = None if k == 0 else Lg["iters"][k - 1]["P"]
        o2 = orc.iteration(o_ref, o_cur, level, ig["T"], ping, orc.RCP_EXACT, *sel)
        assert o2["n"] == ig["V"], (where, "constraint counts at the GPU's pose", o2["n"], ig["V"])
        prior_g = mu * np.asarray(orc.se3_log(ig["initial"])) if mu else np.zeros(6)
        x2 = np.linalg.solve(np.asarray(o2["A"], np.float64) + mu * np.eye(6), np.asarray(o2["b"], np.float64) + prior_g)
        _, _, band2 = _noise_band(orc, o_ref, o_cur, level, ig["T"], ping, o2["ll"], x_ref=x2, mu=mu, prior=prior_g, sel=sel)
        assert np.abs(x2 - ig["inc"]).max() <= 2 * band2 + 1e-12, (where, "GPU increment", ig["inc"], "reference arithmetic", x2, band2)
        assert (np.abs(x2).max() > precision) == cont_g or abs(np.abs(x2).max() - precision) <= band2, \
            (where, "at the GPU's pose the reference arithmetic gives |x|_inf", np.abs(x2).max(), "GPU continued:", cont_g)
        report.append(f"{where}: stop / continue flipped by POSE DRIFT -- at the GPU's own pose the reference arithmetic gives "
                      f"|x|_inf = {np.abs(x2).max():.3g} (Precision {precision:g}, band {band2:.3g}) and decides like the GPU")
        return report
    raise AssertionError((where, "a fork that is neither an accept / reject nor a stop / continue flip",
                          ig["V"], io["V"], ig["has_inc"], io["has_inc"]))
