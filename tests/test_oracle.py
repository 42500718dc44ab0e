"""CPU tests of the oracle (oracle/dvo_oracle.c): the reference ships no tests or fixtures for this path, so the
restatement is pinned by textbook identities, finite differences, hand-computed small images, a second (pure numpy,
scalar-loop) restatement of the order-dependent quirks, and the committed golden vectors it generated."""
import ctypes as C
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
f32 = np.float32


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


# ---------------------------------------------------------------------------------------------------------------------
# SE(3): Sophus SE3d closed forms (dense_tracking.cpp:147,238,259-263,371)
# ---------------------------------------------------------------------------------------------------------------------
def test_se3_exp_log_roundtrip(orc, synth):
    rng = np.random.default_rng(1)
    for scale in (1e-12, 1e-6, 1e-2, 0.5, 1.0):  # |omega| stays below pi
        for _ in range(5):
            xi = rng.normal(size=6) * scale
            T = orc.se3_exp(xi)
            assert np.allclose(T[:3, :3] @ T[:3, :3].T, np.eye(3), atol=1e-12)
            assert abs(np.linalg.det(T[:3, :3]) - 1) < 1e-12
            assert np.allclose(orc.se3_log(T), xi, rtol=1e-9, atol=1e-14)
            assert np.allclose(T, synth.se3_exp(xi), atol=1e-12)  # independent numpy implementation
    assert np.array_equal(orc.se3_exp(np.zeros(6)), np.eye(4))


def test_se3_tangent_order_is_upsilon_omega(orc):
    T = orc.se3_exp(np.array([1.0, 2.0, 3.0, 0, 0, 0]))
    assert np.allclose(T[:3, 3], [1, 2, 3]) and np.allclose(T[:3, :3], np.eye(3))
    T = orc.se3_exp(np.array([0, 0, 0, 0, 0, np.pi / 2]))
    assert np.allclose(T[:3, :3], [[0, -1, 0], [1, 0, 0], [0, 0, 1]], atol=1e-12)


# ---------------------------------------------------------------------------------------------------------------------
# Jacobians (dense_tracking.cpp:448-476) against finite differences of pi(exp(xi) p)
# ---------------------------------------------------------------------------------------------------------------------
def test_jacobian_matches_finite_differences(orc, synth):
    rng = np.random.default_rng(2)
    for _ in range(10):
        p = np.array([rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(0.8, 4.0)])
        pf = p.astype(f32)
        Jw = np.zeros(12, f32)
        Jz = np.zeros(6, f32)
        orc.lib().orc_jacobian(_fp(pf), _fp(Jw), _fp(Jz))
        Jw = Jw.reshape(2, 6)
        num = np.zeros((3, 6))
        h = 1e-6
        for k in range(6):
            d = np.zeros(6)
            d[k] = h
            qp = synth.se3_exp(d) @ np.append(pf.astype(np.float64), 1.0)
            qm = synth.se3_exp(-d) @ np.append(pf.astype(np.float64), 1.0)
            num[:2, k] = (qp[:2] / qp[2] - qm[:2] / qm[2]) / (2 * h)
            num[2, k] = (qp[2] - qm[2]) / (2 * h)
        assert np.allclose(Jw, num[:2], rtol=2e-5, atol=2e-5)
        assert np.allclose(Jz, num[2], rtol=2e-5, atol=2e-5)


# ---------------------------------------------------------------------------------------------------------------------
# packed 6x6 accumulator (math_sse.cpp:82-207) against the dense J^T W J
# ---------------------------------------------------------------------------------------------------------------------
def test_rank_update_packing(orc):
    rng = np.random.default_rng(3)
    n = 500
    J = rng.normal(size=(n, 2, 6)).astype(f32)  # row a / row b
    W = np.zeros((n, 2, 2), f32)
    for i in range(n):
        m = rng.normal(size=(2, 2))
        W[i] = (m @ m.T).astype(f32)
    Jcm = np.ascontiguousarray(J.transpose(0, 2, 1))  # column-major 2x6 = [k][a/b]
    Wcm = np.ascontiguousarray(W.transpose(0, 2, 1))
    A = np.zeros(36, f32)
    orc.lib().orc_rank_update(_fp(Jcm), _fp(Wcm), n, _fp(A))
    A = A.reshape(6, 6).T
    ref = sum(J[i].astype(np.float64).T @ W[i].astype(np.float64) @ J[i].astype(np.float64) for i in range(n))
    assert np.array_equal(A, A.T)
    assert np.allclose(A, ref, rtol=1e-4, atol=1e-3)


# ---------------------------------------------------------------------------------------------------------------------
# pyramid / derivatives / selection on a hand-checked 8x8 image (rgbd_image.cpp, point_selection.cpp)
# ---------------------------------------------------------------------------------------------------------------------
def test_pyramid_and_derivatives_8x8(orc):
    I = np.arange(64, dtype=f32).reshape(8, 8) * f32(1.5)
    Z = (f32(1.0) + np.arange(64, dtype=f32).reshape(8, 8) * f32(0.01)).astype(f32)
    Z[2, 3] = np.nan
    p = orc.Pyramid(I, Z, (100.0, 110.0, 3.5, 4.25), 2)
    assert p.size(0) == (8, 8) and p.size(1) == (4, 4)
    assert np.array_equal(p.intrinsics(1), np.array([50.0, 55.0, 1.75, 2.125], f32))
    # level 1: mean of each 2x2 block for intensity, top-left sample for depth
    I1 = (I[0::2, 0::2] + I[0::2, 1::2] + I[1::2, 0::2] + I[1::2, 1::2]) / f32(4.0)
    assert np.array_equal(p.plane(1, 0), I1)
    assert np.array_equal(p.plane(1, 1), Z[0::2, 0::2], equal_nan=True)
    # central differences with clamped borders: interior 0.5*(next-prev), border half the one-sided difference
    Ix = p.plane(0, 2)
    assert np.array_equal(Ix[:, 1:-1], (I[:, 2:] - I[:, :-2]) * f32(0.5))
    assert np.array_equal(Ix[:, 0], (I[:, 1] - I[:, 0]) * f32(0.5))
    assert np.array_equal(Ix[:, -1], (I[:, -1] - I[:, -2]) * f32(0.5))
    Iy = p.plane(0, 3)
    assert np.array_equal(Iy[1:-1], (I[2:] - I[:-2]) * f32(0.5))
    assert np.array_equal(Iy[0], (I[1] - I[0]) * f32(0.5))
    # a NaN depth poisons the depth derivative of its 4 neighbours, not its own (prev/next skip the centre)
    Zx, Zy = p.plane(0, 4), p.plane(0, 5)
    assert np.isnan(Zx[2, 2]) and np.isnan(Zx[2, 4]) and not np.isnan(Zx[2, 3])
    assert np.isnan(Zy[1, 3]) and np.isnan(Zy[3, 3]) and not np.isnan(Zy[2, 3])
    # selection: valid z / zdx / zdy and some non-zero gradient -> everything but the NaN pixel and its 4 neighbours
    rec, idx = p.select(0)
    expect = np.ones((8, 8), bool)
    for (y, x) in [(2, 3), (2, 2), (2, 4), (1, 3), (3, 3)]:
        expect[y, x] = False
    assert np.array_equal(np.sort(idx), np.flatnonzero(expect.ravel()))
    # record = {x,y,z,1, I,Z,Ix,Iy,Zx,Zy,0,0} with x = ((float)u - ox)/fx * z
    k = int(np.flatnonzero(idx == 5 * 8 + 6)[0])
    z = Z[5, 6]
    assert rec[k, 2] == z and rec[k, 3] == 1.0
    assert rec[k, 0] == f32((f32(6.0) - f32(3.5)) / f32(100.0)) * z
    assert rec[k, 1] == f32((f32(5.0) - f32(4.25)) / f32(110.0)) * z
    assert np.array_equal(rec[k, 4:10], [I[5, 6], Z[5, 6], Ix[5, 6], Iy[5, 6], Zx[5, 6], Zy[5, 6]])
    assert np.array_equal(rec[k, 10:], [0, 0])


def test_selection_thresholds_and_flat_image(orc):
    I = np.full((8, 8), 100.0, f32)
    Z = np.full((8, 8), 2.0, f32)
    p = orc.Pyramid(I, Z, (100.0, 100.0, 4.0, 4.0), 1)
    assert p.select(0)[1].size == 0  # no gradient anywhere: '>' with threshold 0 rejects zero derivatives
    I2 = I.copy()
    I2[:, 4:] = 120.0
    p2 = orc.Pyramid(I2, Z, (100.0, 100.0, 4.0, 4.0), 1)
    assert p2.select(0)[1].size == 16  # the two columns next to the step (|Ix| = 10)
    assert p2.select(0, ti=10.0)[1].size == 0 and p2.select(0, ti=9.9)[1].size == 16


# ---------------------------------------------------------------------------------------------------------------------
# residual stage edge cases (dense_tracking_impl.cpp:133-393)
# ---------------------------------------------------------------------------------------------------------------------
def test_residual_stage_edge_cases(orc, small_pair):
    (Ir, Zr), (Ic, Zc), Tgt, K = small_pair
    pr, pc = orc.Pyramid(Ir, Zr, K, 1), orc.Pyramid(Ic, Zc, K, 1)
    n_sel = pr.select(0)[1].size
    pe, r, valid = orc.compute_residuals(pr, pc, 0, np.eye(4))
    assert valid.size == n_sel - (n_sel % 2)  # Q3: an odd trailing point is never processed
    assert r.shape[0] == int(valid.sum()) and 0 < r.shape[0] < n_sel
    # a frame against itself at the identity: photometric and depth residuals vanish up to interpolation round-off
    pe0, r0, _ = orc.compute_residuals(pr, pr, 0, np.eye(4))
    assert np.abs(r0[:, 0]).max() < 1e-3 and np.abs(r0[:, 1]).max() < 1e-3
    # the error record keeps the untransformed reference point
    rec, _ = pr.select(0)
    assert set(map(tuple, pe0[:50, :3])) <= set(map(tuple, rec[:, :3]))
    # everything warped out of the image -> no constraints
    far = np.eye(4)
    far[0, 3] = 50.0
    assert orc.compute_residuals(pr, pc, 0, far)[1].shape[0] == 0
    # occlusion test: current surface 1 m in front of the prediction -> e1 = -1 < -20 sigma_z -> rejected
    pc_near = orc.Pyramid(Ic, Zc - f32(1.0), K, 1)
    assert orc.compute_residuals(pr, pc_near, 0, np.eye(4))[1].shape[0] == 0
    # ... 1 m behind is kept (the test is one-sided)
    pc_far = orc.Pyramid(Ic, Zc + f32(1.0), K, 1)
    assert orc.compute_residuals(pr, pc_far, 0, np.eye(4))[1].shape[0] > 0
    # all-NaN current depth -> no constraints
    pc_nan = orc.Pyramid(Ic, np.full_like(Zc, np.nan), K, 1)
    assert orc.compute_residuals(pr, pc_nan, 0, np.eye(4))[1].shape[0] == 0


# ---------------------------------------------------------------------------------------------------------------------
# weights / scale / log-likelihood: second restatement in scalar numpy float32, incl. Q5, Q6, Q7 tails
# ---------------------------------------------------------------------------------------------------------------------
def _mahal(r, P):
    t0 = f32(f32(r[0] * P[0]) + f32(r[1] * P[1]))
    t1 = f32(f32(r[0] * P[2]) + f32(r[1] * P[3]))
    return f32(f32(t0 * r[0]) + f32(t1 * r[1]))


def _python_stage(res, P_in, unit):
    n = len(res)
    if unit:
        w = np.ones(n, f32)
    else:
        w = np.array([f32(f32(7.0) / f32(f32(5.0) + _mahal(res[i], P_in))) for i in range(n)], f32)
    scale = f32(f32(1.0) / f32(n - 3))
    acc = np.zeros(4, f32)
    n2 = n - n % 2
    for i in range(0, n2, 2):  # Q5: both halves of a pair use the first residual
        x, y = res[i]
        f = [f32(x * x), f32(y * x), f32(x * y), f32(y * y)]
        for k in range(4):
            acc[k] = f32(acc[k] + f32(f32(scale * f32(w[i] * f[k])) + f32(scale * f32(w[i + 1] * f[k]))))
    cov = np.array([acc[0], acc[1], acc[1], acc[3]], f32)
    if n % 2:
        x, y = res[n - 1]
        wx, wy = f32(w[n - 1] * x), f32(w[n - 1] * y)
        cov = cov + np.array([f32(scale * f32(wx * x)), f32(scale * f32(wy * x)), f32(scale * f32(wx * y)),
                              f32(scale * f32(wy * y))], f32)
    det = f32(f32(cov[0] * cov[3]) - f32(cov[1] * cov[2]))
    inv = f32(f32(1.0) / det)
    P = np.array([f32(cov[3] * inv), f32(-cov[1] * inv), f32(-cov[2] * inv), f32(cov[0] * inv)], f32)
    esum, eacc = 0.0, 1.0
    for i in range(n):
        eacc *= 1.0 + 0.2 * float(_mahal(res[i], P))
        if (i + 1) % 50 == 0:  # Q6: the last n % 50 residuals never reach the sum
            esum += np.log(eacc)
            eacc = 1.0
    detp = f32(f32(P[0] * P[3]) - f32(P[1] * P[2]))
    ll = f32(0.5 * n * float(np.log(detp, dtype=f32)) - 3.5 * esum)
    return w, cov, P, ll


@pytest.mark.parametrize("n", [6, 7, 49, 50, 51, 101, 202, 203])
@pytest.mark.parametrize("unit", [True, False])
def test_weights_scale_loglik_against_numpy_restatement(orc, n, unit):
    rng = np.random.default_rng(100 + n)
    res = (rng.normal(size=(n, 2)) * np.array([0.02, 0.05])).astype(f32)
    res[rng.integers(0, n)] *= f32(30.0)  # an outlier
    P_in = np.array([2000.0, -30.0, -30.0, 400.0], f32)
    w = np.zeros(n, f32)
    cov = np.zeros(4, f32)
    P = np.zeros(4, f32)
    ll = orc.lib().orc_weights_scale_loglik(_fp(res), n, _fp(P_in), int(unit), orc.RCP_EXACT, _fp(w), _fp(cov), _fp(P))
    pw, pcov, pP, pll = _python_stage(res, P_in, unit)
    assert np.array_equal(w, pw)
    assert np.array_equal(cov, pcov)
    assert np.array_equal(P, pP)
    assert abs(ll - pll) <= 4 * np.spacing(f32(abs(pll)))  # libm log vs numpy log: a few ulp of the float result


@pytest.mark.parametrize("n", [7, 50, 51, 203])
def test_clean_mode_uses_every_residual_and_the_whole_likelihood(orc, n):
    """ORC_RCP_CLEAN (SURVEY.md's CLEAN oracle): the scale estimate is the plain weighted scatter (no Q5 pairing quirk) and
    the log-likelihood includes the last n % 50 residuals (no Q6); everything else is the exact-reciprocal mode."""
    rng = np.random.default_rng(300 + n)
    res = (rng.normal(size=(n, 2)) * np.array([0.02, 0.05])).astype(f32)
    P_in = np.array([2000.0, -30.0, -30.0, 400.0], f32)
    out = {}
    for mode in (orc.RCP_EXACT, orc.RCP_CLEAN):
        w, cov, P = np.zeros(n, f32), np.zeros(4, f32), np.zeros(4, f32)
        ll = orc.lib().orc_weights_scale_loglik(_fp(res), n, _fp(P_in), 0, mode, _fp(w), _fp(cov), _fp(P))
        out[mode] = (w.copy(), cov.copy(), P.copy(), float(ll))
    w = out[orc.RCP_CLEAN][0]
    assert np.array_equal(w, out[orc.RCP_EXACT][0])  # the weights are the same
    want = (w[:, None, None].astype(np.float64) * res[:, :, None] * res[:, None, :]).sum(0) / (n - 3)
    assert np.allclose(out[orc.RCP_CLEAN][1].reshape(2, 2), want, rtol=1e-5)
    assert not np.allclose(out[orc.RCP_EXACT][1].reshape(2, 2), want, rtol=1e-3)  # Q5 is visible at this size
    P = out[orc.RCP_CLEAN][2].astype(np.float64).reshape(2, 2)
    q = np.einsum("ni,ij,nj->n", res.astype(np.float64), P, res.astype(np.float64))
    ll_full = 0.5 * n * np.log(np.linalg.det(P)) - 3.5 * np.log1p(0.2 * q).sum()
    assert abs(out[orc.RCP_CLEAN][3] - ll_full) <= 2e-4 * abs(ll_full)


def test_sse_rcp_mode_is_a_12bit_reciprocal(orc):
    xs = np.linspace(0.5, 7.9, 1000).astype(f32)
    rel = [abs(orc.lib().orc_host_rcp(float(x)) * float(x) - 1.0) for x in xs]
    assert max(rel) <= 1.5 * 2.0 ** -12 + 1e-7  # Intel/AMD bound for rcpps
    assert max(rel) > 1e-6  # ... and it is not an exact division (Q8 is real on this host)


# ---------------------------------------------------------------------------------------------------------------------
# the driver (dense_tracking.cpp:131-376)
# ---------------------------------------------------------------------------------------------------------------------
def test_match_converges_to_ground_truth(orc, synth, small_pair):
    (Ir, Zr), (Ic, Zc), Tgt, K = small_pair
    pr, pc = orc.Pyramid(Ir, Zr, K, 3), orc.Pyramid(Ic, Zc, K, 3)
    out = {}
    for mode in (orc.RCP_SSE, orc.RCP_EXACT):
        r = orc.match(orc.default_config(first_level=2, last_level=0, rcp_mode=mode), pr, pc)
        out[mode] = r
        assert not r["is_nan"]
        assert synth.pose_error(Tgt, r["T"]) < 5e-4
        assert [L["id"] for L in r["levels"]] == [2, 1, 0]
        for L in r["levels"]:
            assert L["termination"] in (0, 1, 2) and len(L["iterations"]) >= 1
            assert L["max_valid_pixels"] == 160 * 120 // 4 ** L["id"]
            assert L["iterations"][0]["valid_constraints"] <= L["valid_pixels"]
        last = r["levels"][-1]
        its = last["iterations"]
        src = its[-2] if last["termination"] == 2 else its[-1]
        assert np.allclose(r["information"], src["information"] * 0.008 * 0.008)
        assert r["loglik"] == src["tdist_loglik"] + src["prior_loglik"]
    assert synth.pose_error(out[orc.RCP_SSE]["T"], out[orc.RCP_EXACT]["T"]) < 5e-4


def test_match_is_invariant_to_swapping_roles(orc, synth, small_pair):
    (Ir, Zr), (Ic, Zc), Tgt, K = small_pair
    pr, pc = orc.Pyramid(Ir, Zr, K, 3), orc.Pyramid(Ic, Zc, K, 3)
    cfg = orc.default_config(first_level=2, last_level=0, rcp_mode=orc.RCP_EXACT)
    fwd, bwd = orc.match(cfg, pr, pc), orc.match(cfg, pc, pr)
    assert synth.pose_error(np.eye(4), fwd["T"] @ bwd["T"]) < 1e-3


def test_reference_algorithm_is_chaotic(orc, synth):
    """Perturbing the initial pose by 1e-9 (far below any tolerance) changes the reference algorithm's own answer by
    orders of magnitude more: quirks Q5 / Q6 turn a +-1 change of a valid count into a ~1e4 jump of the log-likelihood,
    which flips accept / reject decisions.  This bounds what any re-implementation with a different summation order can
    promise on every input (see tests/fork_criterion.py)."""
    (Ir, Zr), (Ic, Zc), Tgt = synth.make_pair(320, 240)
    K = synth.intrinsics_for(320, 240)
    pr, pc = orc.Pyramid(Ir, Zr, K, 3), orc.Pyramid(Ic, Zc, K, 3)
    cfg = orc.default_config(first_level=2, last_level=0, rcp_mode=orc.RCP_EXACT, use_initial_estimate=1)
    base = orc.match(cfg, pr, pc, np.eye(4))
    worst = 0.0
    jumps = 0
    for k in range(12):
        xi = np.zeros(6)
        xi[k % 6] = 1e-9 * (1 + k)
        r = orc.match(cfg, pr, pc, synth.se3_exp(xi))
        worst = max(worst, synth.pose_error(base["T"], r["T"]))
        for La, Lb in zip(base["levels"], r["levels"]):
            for ia, ib in zip(La["iterations"], Lb["iterations"]):
                if ia["valid_constraints"] != ib["valid_constraints"] and abs(ia["tdist_loglik"] - ib["tdist_loglik"]) > 50:
                    jumps += 1
    assert worst < 1e-3       # still the same basin
    assert worst > 1e-8       # but three orders of magnitude above the perturbation ...
    assert jumps > 0          # ... because a changed valid count moves the likelihood by far more than one term


def test_reference_algorithm_moves_under_reassociated_sums(orc, synth, capsys):
    """The same fp32 terms of the scale estimate and of the normal equations, added up in another order (orc_config.sum_mode:
    an fp64 accumulator, blocked fp32 partial sums -- what ANY implementation does that does not visit the pixels strictly one
    after the other): the reference algorithm's own answer moves, on some inputs by far more than 1e-5, because a last-bit
    difference of a sum flips an accept / reject decision (dense_tracking.cpp:304-322).  This self-distance is the bound
    tests/fork_criterion.py gives a forked GPU run; the configuration below is the one whose GPU result is 5.5e-5 from the
    oracle (tests/test_gpu_parity.py::test_match_swapped_roles_and_larger_motion)."""
    import fork_criterion

    (Ir, Zr), (Ic, Zc), _ = synth.make_pair(640, 480)
    K = synth.intrinsics_for(640, 480)
    ref = orc.Pyramid(Ir, Zr, K, 4)
    out = {}
    for name, cur_frame in (("headline pair", (Ic, Zc)),
                            ("larger motion", synth.render(640, 480, synth.se3_exp([0.04, -0.02, 0.03, 0.015, -0.02, 0.01]), frame_id=5))):
        cur = orc.Pyramid(cur_frame[0], cur_frame[1], K, 4)
        ocfg = orc.default_config(first_level=3, last_level=0, rcp_mode=orc.RCP_EXACT)
        base = orc.match(ocfg, ref, cur)
        assert ocfg.sum_mode == orc.SUM_REFERENCE
        sd = fork_criterion.self_distance(orc, synth, ocfg, ref, cur, None, base)
        out[name] = sd
        # every run is self-consistent: its iteration counts and terminations follow from its own recorded numbers
        fork_criterion.check_self_consistency(fork_criterion.oracle_levels(base), ocfg.precision, ocfg.max_iterations_per_level, name)
        for d, _ in sd.values():
            assert d < 3e-4  # the same basin
    with capsys.disabled():
        for name, sd in out.items():
            print(f"\n[re-association] {name}: " + "; ".join(f"{k}: {d:.2e} from the reference order, path {path}" for k, (d, path) in sd.items()))
    assert max(d for d, _ in out["headline pair"].values()) < 1e-5   # forks (other iteration counts) but stays inside the bar
    assert max(d for d, _ in out["larger motion"].values()) > 1e-5   # leaves it: no summation order can promise 1e-5 here


def test_termination_semantics(orc, small_pair):
    (Ir, Zr), (Ic, Zc), Tgt, K = small_pair
    pr = orc.Pyramid(Ir, Zr, K, 3)
    # no valid current depth: TooFewConstraints on every level, Result::isNaN
    pc_nan = orc.Pyramid(Ic, np.full_like(Zc, np.nan), K, 3)
    r = orc.match(orc.default_config(first_level=2, last_level=0, rcp_mode=orc.RCP_EXACT), pr, pc_nan)
    assert r["is_nan"] and np.allclose(r["T"], np.eye(4))
    for L in r["levels"]:
        assert len(L["iterations"]) == 1 and L["iterations"][0]["valid_constraints"] == 0
        # x = log(identity) = 0 <= Precision overrides TooFewConstraints after the break (dense_tracking.cpp:359-360)
        assert L["termination"] == 1
    # one iteration per level: IterationsExceeded, the single increment is never applied
    pc = orc.Pyramid(Ic, Zc, K, 3)
    r1 = orc.match(orc.default_config(first_level=2, last_level=0, rcp_mode=orc.RCP_EXACT, max_iterations_per_level=1), pr, pc)
    assert all(L["termination"] == 0 and len(L["iterations"]) == 1 for L in r1["levels"])
    assert not r1["is_nan"]
    # insane config
    with pytest.raises(RuntimeError):
        orc.match(orc.default_config(first_level=0, last_level=1), pr, pc)


def test_initial_estimate_and_prior(orc, synth, small_pair):
    (Ir, Zr), (Ic, Zc), Tgt, K = small_pair
    pr, pc = orc.Pyramid(Ir, Zr, K, 3), orc.Pyramid(Ic, Zc, K, 3)
    cfg = orc.default_config(first_level=2, last_level=0, rcp_mode=orc.RCP_EXACT, use_initial_estimate=1, mu=0.05)
    r = orc.match(cfg, pr, pc, Tgt)
    assert synth.pose_error(Tgt, r["T"]) < 5e-4
    assert r["levels"][0]["iterations"][0]["prior_loglik"] >= 0.0
    # without use_initial_estimate the provided transform is ignored (dense_tracking.cpp:137-144)
    cfg0 = orc.default_config(first_level=2, last_level=0, rcp_mode=orc.RCP_EXACT)
    a, b = orc.match(cfg0, pr, pc, Tgt), orc.match(cfg0, pr, pc, None)
    assert np.array_equal(a["T"], b["T"])


# ---------------------------------------------------------------------------------------------------------------------
# golden vectors (tests/golden/make_golden.py)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["pair_160x120_l3", "pair_320x240_l4_mu"])
def test_oracle_reproduces_golden_vectors(name):
    import importlib.util

    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    got = mg.run_case(name)
    want = np.load(os.path.join(GOLDEN, name + ".npz"))
    assert np.array_equal(got["levels"], want["levels"])
    assert np.array_equal(got["sel_counts"], want["sel_counts"])
    assert got["res_count"] == want["res_count"]
    assert np.array_equal(got["res_sample"], want["res_sample"])
    assert np.allclose(got["plane_sums"], want["plane_sums"], rtol=1e-12)
    assert np.allclose(got["T"], want["T"], atol=1e-12)
    assert np.allclose(got["iterations"], want["iterations"], rtol=1e-9, atol=1e-12)
    assert np.allclose(got["information"], want["information"], rtol=1e-9)


# ---------------------------------------------------------------------------------------------------------------------
# frame ingest
# ---------------------------------------------------------------------------------------------------------------------
def test_ingest_depth_zero_is_nan_and_scale_is_one_fp32_multiply(orc):
    rng = np.random.default_rng(5)
    raw = rng.integers(0, 65536, size=(24, 32), dtype=np.uint16)
    raw[3, 4:9] = 0
    raw[0, 0], raw[0, 1] = 65535, 1
    for scale in (1.0 / 5000.0, 0.001):  # benchmark_slam.cpp:77, camera_dense_tracking.cpp:234
        out = orc.ingest_depth(raw, scale)
        expect = raw.astype(np.float32) * np.float32(scale)
        expect[raw == 0] = np.nan
        assert np.array_equal(out.view(np.uint32)[raw != 0], expect.view(np.uint32)[raw != 0])
        assert np.isnan(out[raw == 0]).all() and np.isfinite(out[raw != 0]).all()


def test_ingest_gray_known_answers_of_the_8bit_rule(orc):
    # the well-known 8-bit luma values of cv::cvtColor(BGR2GRAY): pure blue 29, green 150, red 76; grays are fixed points
    img = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255]],
                    [[0, 0, 0], [128, 128, 128], [17, 17, 17], [10, 200, 90]]], np.uint8)
    out = orc.ingest_gray(img)
    assert out.dtype == np.float32
    assert out.tolist() == [[29.0, 150.0, 76.0, 255.0], [0.0, 128.0, 17.0, float((10 * 1868 + 200 * 9617 + 90 * 4899 + 8192) >> 14)]]
    rng = np.random.default_rng(6)
    img = rng.integers(0, 256, size=(20, 28, 3), dtype=np.uint8)
    b, g, r = [img[..., c].astype(np.int64) for c in range(3)]
    assert np.array_equal(orc.ingest_gray(img), ((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14).astype(np.float32))
    # within half a gray level of the real-valued luma everywhere
    assert np.abs(orc.ingest_gray(img) - (0.114 * b + 0.587 * g + 0.299 * r)).max() <= 0.5 + 0.02  # Q14 coefficients
    gray = rng.integers(0, 256, size=(8, 12), dtype=np.uint8)
    assert np.array_equal(orc.ingest_gray(gray), gray.astype(np.float32))


def test_raw_synthetic_frame_round_trips_through_ingest(orc, synth):
    (I, Z), _, _ = synth.make_pair(64, 48)
    bgr, raw_z = synth.to_raw(I, Z)
    assert bgr.shape == (48, 64, 3) and raw_z.dtype == np.uint16
    Ig, Zg = orc.ingest_gray(bgr), orc.ingest_depth(raw_z)
    assert np.array_equal(np.isnan(Zg), np.isnan(Z))
    m = ~np.isnan(Z)
    assert np.abs(Zg[m] - Z[m]).max() <= 0.5 / 5000.0 + 1e-6
    assert np.abs(Ig - I).max() <= 16.5


def test_iteration_probe_reproduces_the_iterations_of_match(orc, synth, small_pair):
    """orc_iteration (the stage oracle of the weighted-iteration GPU tests) is the loop body of orc_match: fed the pose and
    the previous precision of iterations 0 and 1 of a level it reproduces their statistics bit for bit."""
    (Ir, Zr), (Ic, Zc), Tgt, K = small_pair
    pr, pc = orc.Pyramid(Ir, Zr, K, 2), orc.Pyramid(Ic, Zc, K, 2)
    m = orc.match(orc.default_config(first_level=1, last_level=1, rcp_mode=orc.RCP_EXACT), pr, pc)
    its = m["levels"][0]["iterations"]
    assert len(its) >= 3
    T, prec = np.eye(4), None
    for k in range(2):
        it = orc.iteration(pr, pc, 1, T, prec, orc.RCP_EXACT)
        assert it["n"] == its[k]["valid_constraints"]
        assert np.array_equal(it["precision"].astype(np.float64), its[k]["precision"])
        assert np.float64(it["ll"]) == -its[k]["tdist_loglik"]
        assert np.array_equal(it["A"].astype(np.float64), its[k]["information"])
        assert np.array_equal(it["b"].astype(np.float64), its[k]["rhs"])
        T = orc.se3_exp(its[k]["increment"]) @ T  # estimate = inc * estimate (dense_tracking.cpp:261)
        prec = it["precision"]


def test_float64_restatement_brackets_the_oracles_sequential_fp32_sums(orc, synth):
    """tests/stage_f64.py restates one iteration body a third time (numpy, float64 sums, on the oracle's bit-exact residual
    records).  The oracle -- like the reference -- accumulates scale, A and b sequentially in fp32; this pins how far that
    alone is from the exact sums (the error bars the GPU stage tests grant the reference, tests/test_gpu_parity.py REF_*)."""
    from stage_f64 import f64_iteration

    (Ir, Zr), (Ic, Zc), Tgt = synth.make_pair(320, 240)
    K = synth.intrinsics_for(320, 240)
    pr, pc = orc.Pyramid(Ir, Zr, K, 3), orc.Pyramid(Ic, Zc, K, 3)
    ro = orc.match(orc.default_config(first_level=2, last_level=0, rcp_mode=orc.RCP_EXACT), pr, pc)
    worst = dict(scale=0.0, A=0.0, b=0.0, ll=0.0)
    for L in ro["levels"]:
        prec = None
        for it in L["iterations"]:
            n, cov, A, b, cs, ll = f64_iteration(orc, pr, pc, L["id"], it["estimate"], prec, it["precision"])
            assert n == it["valid_constraints"]
            worst["scale"] = max(worst["scale"], np.abs(cov - it["scale"]).max() / np.abs(cov).max())
            worst["ll"] = max(worst["ll"], abs(-ll - it["tdist_loglik"]) / abs(ll))
            if it["has_increment"]:
                worst["A"] = max(worst["A"], np.abs(A - it["information"]).max() / np.abs(A).max())
                worst["b"] = max(worst["b"], (np.abs(b - it["rhs"]) / cs).max())
            prec = it["precision"]
    assert worst["scale"] <= 1e-3 and worst["A"] <= 3e-4 and worst["b"] <= 3e-5 and worst["ll"] <= 2e-6, worst
    assert worst["scale"] >= 1e-7  # it is an fp32 sum: if this ever reads 0 the comparison has stopped comparing


# ---------------------------------------------------------------------------------------------------------------------
# the continuation entry (orc_match_from) and the deterministic fork criterion built on it (tests/fork_criterion.py)
# ---------------------------------------------------------------------------------------------------------------------
def _path(r):
    return [(L["id"], L["termination"], len(L["iterations"])) for L in r["levels"]]


def test_continuation_reproduces_the_rest_of_a_match(orc, synth):
    """orc_match_from entered with the state orc_match itself held at the top of ANY iteration body -- inside a level or at a
    level start -- runs the rest of that match: same iterations, same terminations, same numbers, same result
    (dense_tracking.cpp:247-363 and the level loop around it hold no other state)."""
    import fork_criterion as F

    (Ir, Zr), (Ic, Zc), _ = synth.make_pair(320, 240)
    K = synth.intrinsics_for(320, 240)
    pr, pc = orc.Pyramid(Ir, Zr, K, 3), orc.Pyramid(Ic, Zc, K, 3)
    n_states = 0
    for kw, T0 in ((dict(), None), (dict(mu=0.05, use_initial_estimate=1, precision=1e-4), synth.se3_exp([0.01, 0, -0.01, 0, 0.004, 0]))):
        cfg = orc.default_config(first_level=2, last_level=0, rcp_mode=orc.RCP_EXACT, **kw)
        base = orc.match(cfg, pr, pc, T0)
        O = F.oracle_levels(base)
        for li, L in enumerate(O):
            for k in range(len(L["iters"])):
                st = F.state_behind(orc, O, cfg, T0, li, k)
                if st is None:
                    continue
                rest = orc.match_from(cfg, pr, pc, **st)
                n_states += 1
                want = _path(base)[li + 1:] if st["iteration"] == 0 else \
                    [(L["id"], L["termination"], len(L["iters"]) - k - 1)] + _path(base)[li + 1:]
                assert _path(rest) == want, (li, k, _path(rest), want)
                assert synth.pose_error(base["T"], rest["T"]) < 1e-13
                assert np.allclose(rest["information"], base["information"], rtol=1e-12, equal_nan=True)
                assert rest["loglik"] == base["loglik"] or (np.isnan(rest["loglik"]) and np.isnan(base["loglik"]))
                # every continued iteration is the original's, number for number
                spliced = F.splice(O, li, k, st, F.oracle_levels(rest))
                for La, Lb in zip(spliced, O):
                    assert len(La["iters"]) == len(Lb["iters"])
                    for ia, ib in zip(La["iters"], Lb["iters"]):
                        assert ia["V"] == ib["V"] and ia["nll"] == ib["nll"] and ia["has_inc"] == ib["has_inc"]
                        if ia["has_inc"]:
                            assert np.allclose(ia["inc"], ib["inc"], rtol=1e-9, atol=1e-15)
    assert n_states >= 20
    # a state the reference cannot be in is refused
    with pytest.raises(RuntimeError):
        orc.match_from(cfg, pr, pc, level=5, iteration=0, estimate=np.eye(4), initial=np.eye(4), x=np.zeros(6))


class _StandIn:
    """an oracle run in capi.Result's shape: stands in for the GPU where there is none (the fork criterion on the CPU)"""

    def __init__(self, r):
        self.Transformation, self.Information, self.LogLikelihood = r["T"], r["information"], r["loglik"]
        self.Levels = [{"Id": L["id"], "TerminationCriterion": L["termination"], "ValidPixels": L["valid_pixels"],
                        "MaxValidPixels": L["max_valid_pixels"],
                        "Iterations": [{"ValidConstraints": it["valid_constraints"], "TDistributionLogLikelihood": it["tdist_loglik"],
                                        "has_increment": it["has_increment"], "EstimateIncrement": it["increment"],
                                        "TDistributionPrecision": it["precision"], "estimate": it["estimate"], "initial": it["initial"],
                                        "EstimateInformation": it["information"], "PriorLogLikelihood": it["prior_loglik"]}
                                       for it in L["iterations"]]} for L in r["levels"]]

    def isNaN(self):
        return False


def test_fork_criterion_resynchronises_a_stand_in(orc, synth, capsys):
    """The criterion of tests/fork_criterion.py end to end on the CPU.  The stand-in for the GPU is the oracle with its fp32 terms
    accumulated in double (another summation order, like any GPU): on the "larger motion" pair it forks from the reference-order
    oracle and lands 5.5e-5 from it -- beyond the 1e-5 bar, which rounds 3-4 excused with a sampled self-distance.  Now: the
    flipped decision is adjudicated, the reference-order oracle is continued from the stand-in's own state behind it, the rest is
    same-path and the final pose is within 1e-5 of the continuation's.  And a stand-in whose pose bookkeeping is WRONG behind the
    fork is caught."""
    import copy

    import fork_criterion as F

    (Ir, Zr), _, _ = synth.make_pair(640, 480)
    K = synth.intrinsics_for(640, 480)
    ref = orc.Pyramid(Ir, Zr, K, 4)
    cur_frame = synth.render(640, 480, synth.se3_exp([0.04, -0.02, 0.03, 0.015, -0.02, 0.01]), frame_id=5)
    cur = orc.Pyramid(cur_frame[0], cur_frame[1], K, 4)
    ocfg = orc.default_config(first_level=3, last_level=0, rcp_mode=orc.RCP_EXACT)
    ro = orc.match(ocfg, ref, cur)
    lines = []
    n_forked = 0
    for mode in (orc.SUM_FP64, orc.SUM_BLOCKED):
        other = orc.match(orc.default_config(first_level=3, last_level=0, rcp_mode=orc.RCP_EXACT, sum_mode=mode), ref, cur)
        g = _StandIn(other)
        err = synth.pose_error(ro["T"], g.Transformation)
        if F.same_path(g, ro):
            continue
        n_forked += 1
        report, O, final = F.adjudicate(orc, synth, ocfg, ref, cur, None, g, ro, err, 1e-5)
        assert F.first_fork(F.gpu_levels(g), O) is None
        assert synth.pose_error(final["T"], g.Transformation) <= 1e-5 and np.allclose(final["information"], g.Information, rtol=5e-3)
        lines += [f"[stand-in sum_mode {mode}] {ln}" for ln in report]
        if err > 1e-5:
            # a run that is NOT the reference algorithm behind the fork: its last level's estimates moved by 3e-5
            bad = copy.deepcopy(g)
            shift = synth.se3_exp([3e-5, 0, 0, 0, 0, 0])
            bad.Transformation = bad.Transformation @ shift
            with pytest.raises(AssertionError):
                F.adjudicate(orc, synth, ocfg, ref, cur, None, bad, ro, err, 1e-5)
    assert n_forked >= 1
    with capsys.disabled():
        print()
        for ln in lines:
            print(ln)


def test_q7_tail_weights_are_below_every_sums_rounding(orc, synth, capsys):
    """Q7: computeWeightsSse forms the first 4 * floor(V / 4) weights with rcpps and the last V mod 4 with an exact division
    (dense_tracking_impl.cpp:667-706).  The HIP path's host-rcpps mode takes EVERY weight from the rcpps table -- a stated deviation
    (include/dvo_amd.h, DVO_AMD_RCP_HOST_SSE).  Its size, measured on the reference's own arithmetic: the at most three tail weights
    change by <= 2^-12 relative, i.e. the weighted sums a weight enters (scale, A, b) by <= 3 / V * 2^-12 ~ 4e-9 of their size at
    V = 2e5 -- two orders of magnitude below the fp32 rounding of a single term and five below the 2e-4 by which the reference's own
    sequential fp32 sums miss the exact ones.  No test at the level of a sum, an increment or a pose can see it."""
    import ctypes as C

    (Ir, Zr), (Ic, Zc), Tgt = synth.make_pair(640, 480)
    K = synth.intrinsics_for(640, 480)
    pr, pc = orc.Pyramid(Ir, Zr, K, 1), orc.Pyramid(Ic, Zc, K, 1)
    pe, res, _ = orc.compute_residuals(pr, pc, 0, Tgt, orc.RCP_SSE)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
    worst = 0.0
    for tail in range(4):  # V mod 4 = 0 .. 3
        r = np.ascontiguousarray(res[: len(res) - ((len(res) - tail) % 4)], np.float32)
        n = len(r)
        assert n % 4 == tail
        P0 = np.array([1e-3, 0, 0, 1e5], np.float32)  # a previous precision of the size the level-0 iterations see
        w_ref, cov, P = np.zeros(n, np.float32), np.zeros(4, np.float32), np.zeros(4, np.float32)
        orc.lib().orc_weights_scale_loglik(fp(r), n, fp(P0), 0, orc.RCP_SSE, fp(w_ref), fp(cov), fp(P))
        d = (r[:, 0] * P0[0] + r[:, 1] * P0[1]) * r[:, 0] + (r[:, 0] * P0[2] + r[:, 1] * P0[3]) * r[:, 1]
        w_all = (np.float32(7.0) * orc.host_rcp((np.float32(5.0) + d).astype(np.float32))).astype(np.float32)
        n4 = n - n % 4
        assert np.array_equal(w_all[:n4], w_ref[:n4])            # the body: the very rcpps weights
        tail_rel = np.abs(w_all[n4:].astype(np.float64) - w_ref[n4:]) / w_ref[n4:] if n % 4 else np.zeros(1)
        assert tail_rel.max() <= 2.0 ** -11
        # what that does to a weighted sum of the pass (float64, so that nothing else is in the difference)
        rr = (r[:, 0].astype(np.float64) ** 2)
        s_ref, s_all = (w_ref.astype(np.float64) * rr).sum(), (w_all.astype(np.float64) * rr).sum()
        worst = max(worst, abs(s_all - s_ref) / abs(s_ref))
    with capsys.disabled():
        print(f"\n[Q7 tail] all-table weights vs the reference's exact-division tail: a weighted sum of the pass moves by at most {worst:.1e} relative")
    assert worst < 1e-7
