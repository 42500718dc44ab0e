"""The opt-in reciprocal mode that reproduces the HOST's _mm_rcp_ps (SURVEY Q8 and the body of Q7).

The reference forms 1 / z of the projection (dvo_core/src/dense_tracking_impl.cpp:192) and the t-distribution weights (:700)
with rcpps, a ~12-bit approximation whose bits differ between CPU vendors.  By default the HIP path uses the exactly truncated
quotient (the same on every machine); with dvo_amd_set_reciprocal_mode(DVO_AMD_RCP_HOST_SSE) it uses a table of rcpps(1.m)
probed on the host it runs on.  Checked here against the oracle's RCP_SSE mode, which executes the real instruction:

  * the table reciprocal == _mm_rcp_ps on this host, bit for bit, for every class of input;
  * residuals and validity decisions bit-exact at every level;
  * every t-distribution weight of a pass bit-exact, the exact-division tail (Q7) included;
  * match(): same-path poses <= 1e-5, forks under the self-distance rule; the three flavours of "the reference SSE path"
    (portable oracle with rcpps, natively built FMA-contracting oracle with rcpps, exact-reciprocal oracle) on one line.
"""
import numpy as np
import pytest

import fork_criterion
import test_gpu_parity as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from dvo_slam_amd import capi as c

    if c.lib().dvo_amd_device_count() < 1:
        pytest.fail("no HIP device visible: the gpu tests must run on the MI355X box")
    return c


@pytest.fixture(scope="module")
def pair(capi, orc, synth):
    (Ir, Zr), (Ic, Zc), Tgt = synth.make_pair(640, 480)
    K = synth.intrinsics_for(640, 480)
    return dict(gr=capi.RgbdImagePyramid(Ir, Zr, K, 4), gc=capi.RgbdImagePyramid(Ic, Zc, K, 4),
                orr=orc.Pyramid(Ir, Zr, K, 4), occ=orc.Pyramid(Ic, Zc, K, 4), Tgt=Tgt, K=K, frames=((Ir, Zr), (Ic, Zc)))


def _tracker(capi, **cfg):
    trk = capi.DenseTracker(capi.Config(**cfg))
    trk.set_reciprocal_mode("host_sse")
    return trk


def test_table_reciprocal_is_the_hosts_rcpps_bit_for_bit(capi, orc):
    trk = _tracker(capi, FirstLevel=3, LastLevel=0)
    mode, k = trk.reciprocal_mode()
    assert mode == "host_sse" and 8 <= k <= 23
    print(f"\n[rcpps] this host's _mm_rcp_ps depends on the top {k} mantissa bits: a table of {1 << k} entries ({(4 << k) / 1024:.0f} KiB)")
    rng = np.random.default_rng(20131103)
    cases = [rng.integers(0, 2**32, size=2_000_000, dtype=np.uint64).astype(np.uint32),              # every exponent, both signs
             (np.uint32(0x3f800000) + np.arange(1 << 23, dtype=np.uint32)[:: 7]),                    # a dense sweep of [1, 2)
             (np.uint32(0x40000000) + rng.integers(0, 1 << 23, size=300_000).astype(np.uint32)),     # depths of 2 .. 4 m
             np.array([0x00000000, 0x80000000, 0x00000001, 0x007fffff, 0x00800000, 0x807fffff, 0x7e800000, 0x7e7fffff, 0x7e800001,
                       0x7f000000, 0x7f7fffff, 0x7f800000, 0xff800000, 0x7fc00000, 0x7f800001, 0xffc12345, 0x3f800000, 0x3f7fffff,
                       0xbf800000, 0x40a00000], dtype=np.uint32)]
    for bits in cases:
        x = bits.view(np.float32)
        want, got = orc.host_rcp(x).view(np.uint32), trk.table_rcp(x).view(np.uint32)
        nan = np.isnan(x)
        assert np.array_equal(want[~nan], got[~nan])
        assert np.isnan(got.view(np.float32)[nan]).all()  # NaN in, NaN out (the kernels only ask whether it is one)
    # the exact mode is the default and has no table
    plain = capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=0))
    assert plain.reciprocal_mode() == ("exact", 0)
    with pytest.raises(capi.DvoAmdError):
        plain.table_rcp(np.ones(4, np.float32))


def _oracle_residual_image(orc, orr, occ, level, T, shape, mode):
    pe, r, valid = orc.compute_residuals(orr, occ, level, T, mode)
    rec, idx = orr.select(level)
    img = np.full((shape[0] * shape[1], 2), np.nan, np.float32)
    img[idx[: len(valid)][valid.astype(bool)]] = r
    return img.reshape(shape[0], shape[1], 2), len(r)


def test_the_two_forms_of_the_mode_are_the_same_function(capi, orc, synth, pair, monkeypatch, capsys):
    """Round 5: the mode's reciprocal without a global-memory gather -- the device's own v_rcp_f32 of the midpoint of the input's
    table cell plus a signed 4-bit correction from a 2 KiB table in LDS (dvo_kernels.hip: rcp_host_nibbles), the corrections formed
    on the device when the mode is switched on.  Same function as the table form by construction: the instruction bit for bit
    (the test above runs on whichever form the tracker picked), and a whole match() -- every statistic -- identical to the table
    form's (DVO_AMD_RCP_FORM=table keeps the round-4 form)."""
    nib = _tracker(capi, FirstLevel=3, LastLevel=0)
    form, why = nib.reciprocal_form()
    with capsys.disabled():
        print(f"\n[rcpps] form in use: {form}" + (f" ({why})" if why else ""))
    monkeypatch.setenv("DVO_AMD_RCP_FORM", "table")
    tab = _tracker(capi, FirstLevel=3, LastLevel=0)
    assert tab.reciprocal_form()[0] == "table"
    monkeypatch.delenv("DVO_AMD_RCP_FORM")
    if form != "nibbles":
        pytest.skip("the nibble form is not available on this host: " + why)
    rng = np.random.default_rng(7)
    x = rng.integers(0, 2**32, size=500_000, dtype=np.uint64).astype(np.uint32).view(np.float32)
    a, b = nib.table_rcp(x).view(np.uint32), tab.table_rcp(x).view(np.uint32)
    ok = ~np.isnan(x)
    assert np.array_equal(a[ok], b[ok])
    for gr, gc in ((pair["gr"], pair["gc"]), (pair["gc"], pair["gr"])):
        ra, rb = nib.match(gr, gc), tab.match(gr, gc)
        assert np.array_equal(ra.Transformation, rb.Transformation) and np.array_equal(ra.Information, rb.Information)
        assert ra.LogLikelihood == rb.LogLikelihood
        assert [[it["ValidConstraints"] for it in L["Iterations"]] for L in ra.Levels] == \
               [[it["ValidConstraints"] for it in L["Iterations"]] for L in rb.Levels]
    out = nib.match_batch([pair["gr"]] * 20, [pair["gc"]] * 20, in_flight=12)
    want = tab.match(pair["gr"], pair["gc"])
    assert all(np.array_equal(want.Transformation, r.Transformation) for r in out)


@pytest.mark.parametrize("level", [3, 2, 1, 0])
def test_residuals_bit_exact_against_the_oracles_rcpps_mode(capi, orc, synth, pair, level):
    trk = _tracker(capi, FirstLevel=3, LastLevel=0)
    odd = synth.se3_exp([0.05, -0.08, 0.1, 0.03, -0.02, 0.04])
    differs_from_exact = 0
    for T in (np.eye(4), pair["Tgt"], np.linalg.inv(pair["Tgt"]), odd):
        g, n_gpu = trk.residuals(pair["gr"], pair["gc"], level, T)
        o, n_orc = _oracle_residual_image(orc, pair["orr"], pair["occ"], level, T, g.shape[:2], orc.RCP_SSE)
        assert n_gpu == n_orc
        assert np.array_equal(np.isnan(g), np.isnan(o))
        m = ~np.isnan(g)
        assert np.array_equal(P._bits(g[m]), P._bits(o[m]))
        e, _ = _oracle_residual_image(orc, pair["orr"], pair["occ"], level, T, g.shape[:2], orc.RCP_EXACT)
        both = m & ~np.isnan(e)
        differs_from_exact += int((P._bits(g[both]) != P._bits(e[both])).sum())
    assert differs_from_exact > 0  # (the two reciprocals are different functions: the mode is really on)


_Q7_SEEN = set()


@pytest.mark.parametrize("level", [3, 2, 1, 0])
def test_every_weight_is_computeWeightsSse_as_this_host_runs_it(capi, orc, synth, pair, level, capsys):
    """computeWeightsSse (dense_tracking_impl.cpp:657-707) forms the first 4 floor(V / 4) weights of a pass as 7 rcpps(5 + d), d
    product by product, and the last V mod 4 by computeWeight's exact division in double (Q7).  In the host-rcpps mode the HIP path
    does the same: the residual pass weights every pixel with the table, k_q7_tail finds the pass's last V mod 4 valid pixels once
    V is known, recomputes them and leaves what their exact weights add to the pair sums and the 87 moments (k_finalize adds it to
    the record).  Checked on the product kernels' own weights (dvo_amd_debug_weights: the residual pass stores them): every weight
    of the body and every tail weight against the oracle's tdist_weights on the same residuals, bit for bit; the tail's pixels;
    the additions against a float64 restatement from the oracle's point records."""
    import ctypes as C

    trk = _tracker(capi, FirstLevel=3, LastLevel=0)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
    poses = [np.eye(4), pair["Tgt"], np.linalg.inv(pair["Tgt"]), synth.se3_exp([0.05, -0.08, 0.1, 0.03, -0.02, 0.04]),
             synth.se3_exp([0.004, 0.002, -0.003, 0.001, -0.002, 0.0015]), synth.se3_exp([-0.01, 0.006, 0.002, -0.003, 0.001, 0.002])]
    for T in poses:
        g, V = trk.residuals(pair["gr"], pair["gc"], level, T)
        m = ~np.isnan(g[..., 0])
        assert int(m.sum()) == V
        res = np.ascontiguousarray(g[m])  # row-major = scan order = the reference's compacted list (bit-exact: the test above)
        # a realistic precision: the one the pass's own unit-weight scale gives
        w_unit, cov, P_in = np.zeros(V, np.float32), np.zeros(4, np.float32), np.zeros(4, np.float32)
        orc.lib().orc_weights_scale_loglik(fp(res), V, fp(P_in), 1, orc.RCP_SSE, fp(w_unit), fp(cov), fp(P_in))
        w_ref, cov2, P2 = np.zeros(V, np.float32), np.zeros(4, np.float32), np.zeros(4, np.float32)
        orc.lib().orc_weights_scale_loglik(fp(res), V, fp(P_in), 0, orc.RCP_SSE, fp(w_ref), fp(cov2), fp(P2))
        w_img, q = trk.weights_probe(pair["gr"], pair["gc"], level, T, P_in.reshape(2, 2).T)
        assert np.array_equal(np.isnan(w_img), ~m)
        w_gpu = w_img[m]
        t = V % 4
        _Q7_SEEN.add(t)
        assert q["n"] == V and q["n_counted"] == V and q["n_tail"] == t and q["recomputed_equal"]
        # the body: rcpps weights, bit for bit
        assert np.array_equal(P._bits(w_gpu[: V - t]), P._bits(w_ref[: V - t]))
        # the tail: the pass gave those pixels the table's weight; k_q7_tail found them and formed computeWeight's
        flat = np.flatnonzero(m.ravel())
        w_all_table = np.float32(7.0) * orc.host_rcp(np.float32(5.0) + _mahalanobis(res, P_in))
        for j in range(t):
            k = V - t + j
            assert q["pixel"][j] == flat[k]
            assert P._bits(q["w_table"][j : j + 1])[0] == P._bits(w_gpu[k : k + 1])[0] == P._bits(w_all_table[k : k + 1])[0]
            assert P._bits(q["w_exact"][j : j + 1])[0] == P._bits(w_ref[k : k + 1])[0]
        assert all(px == -1 for px in q["pixel"][t:])
        # what the exact weights add: pair sums (Q5: an odd rank weights its partner's residual) and moments, in float64
        pe, r_o, _ = orc.compute_residuals(pair["orr"], pair["occ"], level, T, orc.RCP_SSE)
        assert np.array_equal(P._bits(r_o), P._bits(res))
        dS, dM = np.zeros(3), np.zeros(87)
        for j in range(t):
            k = V - t + j
            dw = float(w_ref[k]) - float(w_gpu[k])
            a = res[k] if j % 2 == 0 else res[V - t]
            dS += dw * np.array([float(a[0] * a[0]), float(a[0] * a[1]), float(a[1] * a[1])])
            dM += dw * _moments_of_point(orc, pe[k], res[k])
        assert np.allclose(q["scale_sums_delta"], dS, rtol=1e-12, atol=0.0)
        assert np.allclose(q["moments_delta"], dM, rtol=1e-4, atol=1e-6 * np.abs(dM).max() if t else 0.0)
        if t == 0:
            assert not q["scale_sums_delta"].any() and not q["moments_delta"].any()
    if level == 0:
        with capsys.disabled():
            print(f"\n[Q7] V mod 4 of the passes checked: {sorted(_Q7_SEEN)}")
        assert _Q7_SEEN >= {1, 2, 3}, "add poses: a tail length is not covered"


def _mahalanobis(res, P_colmajor):
    """r^T P r as computeWeightsSse evaluates it (float32, product by product)"""
    r0, r1 = res[:, 0], res[:, 1]
    p = P_colmajor.astype(np.float32)
    t0 = (r0 * p[0]).astype(np.float32) + (r1 * p[1]).astype(np.float32)
    t1 = (r0 * p[2]).astype(np.float32) + (r1 * p[3]).astype(np.float32)
    return ((t0 * r0).astype(np.float32) + (t1 * r1).astype(np.float32)).astype(np.float32)


def _moments_of_point(orc, pe, r):
    """the 87 P-free moments of one point with unit weight (layout: csrc/dvo_types.h kAcc*), from the oracle's point record"""
    import ctypes as C

    Jw, Jz = np.zeros(12, np.float32), np.zeros(6, np.float32)
    p3 = np.ascontiguousarray(pe[:3], np.float32)
    f = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
    orc.lib().orc_jacobian(f(p3), f(Jw), f(Jz))
    Jw, Jz = Jw.astype(np.float64), Jz.astype(np.float64)
    gi0, gi1, gz0, gz1 = (float(x) for x in pe[6:10])  # e[2..5]: the gradient terms (dense_tracking.cpp:333-339)
    Ja = gi0 * Jw[:6] + gi1 * Jw[6:]
    Jb = gz0 * Jw[:6] + gz1 * Jw[6:] - Jz
    out, t = np.zeros(87), 0
    for i in range(6):
        for c in range(i, 6):
            out[t], out[21 + t], out[42 + t] = Ja[i] * Ja[c], Ja[i] * Jb[c] + Jb[i] * Ja[c], Jb[i] * Jb[c]
            t += 1
        out[63 + i], out[69 + i], out[75 + i], out[81 + i] = Ja[i] * r[0], Ja[i] * r[1], Jb[i] * r[0], Jb[i] * r[1]
    return out


def test_match_against_the_three_flavours_of_the_reference_sse_path(capi, orc, synth, pair, capsys):
    """GPU in the host-rcpps mode against the oracle running the real instruction: same path -> 1e-5 (dense_tracking_impl.cpp
    :192,:700), a fork -> adjudicated and re-synchronised (tests/fork_criterion.py).  And on one line: how far the flavours of "the reference SSE path" are from
    each other on this host -- portable build with rcpps, natively built (-O3 -march=native: contracts a*b+c into fma like the
    reference's own flags do, dvo_core/CMakeLists.txt:40-42) with rcpps, exact reciprocal."""
    lines = []
    for label, (gr, gc, orr, occ), cfg in (("headline pair", (pair["gr"], pair["gc"], pair["orr"], pair["occ"]), dict(FirstLevel=3, LastLevel=0)),
                                          ("swapped roles", (pair["gc"], pair["gr"], pair["occ"], pair["orr"]), dict(FirstLevel=3, LastLevel=0)),
                                          ("reference default levels", (pair["gr"], pair["gc"], pair["orr"], pair["occ"]), dict(FirstLevel=3, LastLevel=1))):
        g_sse = _tracker(capi, **cfg).match(gr, gc)
        g_exact = capi.DenseTracker(capi.Config(**cfg)).match(gr, gc)
        ocfg = lambda mode: orc.default_config(first_level=cfg["FirstLevel"], last_level=cfg["LastLevel"], rcp_mode=mode)  # noqa: E731
        o_sse, o_exact = orc.match(ocfg(orc.RCP_SSE), orr, occ), orc.match(ocfg(orc.RCP_EXACT), orr, occ)
        err = synth.pose_error(o_sse["T"], g_sse.Transformation)
        # same path -> 1e-5; forked -> the flipped decision adjudicated and the rcpps oracle continued from the GPU's own state
        # behind it (tests/fork_criterion.py; the reference arithmetic of the criterion runs in the oracle's RCP_SSE mode here)
        same_path, report = fork_criterion.settle(orc, synth, ocfg(orc.RCP_SSE), orr, occ, None, g_sse, o_sse, P.POSE_TOL)
        same_path, note = not same_path, " | ".join(report[1:])
        # first iteration of the first level: identical inputs, identical reciprocal -> identical constraint count
        assert g_sse.Levels[0]["Iterations"][0]["ValidConstraints"] == o_sse["levels"][0]["iterations"][0]["valid_constraints"]
        lines.append(f"[rcpps] {label}: GPU(host rcpps) vs oracle(rcpps) {err:.2e} ({'same path' if same_path else 'forked, ' + (note or 'within 1e-5')}); "
                     f"GPU(exact) vs oracle(exact) {synth.pose_error(o_exact['T'], g_exact.Transformation):.2e}; "
                     f"oracle(rcpps) vs oracle(exact) {synth.pose_error(o_sse['T'], o_exact['T']):.2e}; "
                     f"GPU(exact) vs oracle(rcpps) {synth.pose_error(o_sse['T'], g_exact.Transformation):.2e}")
    # the natively built oracle (fma contraction): objects of one build must not be used under the other
    (Ir, Zr), (Ic, Zc) = pair["frames"]
    try:
        orc.select_build("native")
        n_r, n_c = orc.Pyramid(Ir, Zr, pair["K"], 4), orc.Pyramid(Ic, Zc, pair["K"], 4)
        n_sse = orc.match(orc.default_config(first_level=3, last_level=0, rcp_mode=orc.RCP_SSE), n_r, n_c)
        n_exact = orc.match(orc.default_config(first_level=3, last_level=0, rcp_mode=orc.RCP_EXACT), n_r, n_c)
        del n_r, n_c
    finally:
        orc.select_build("parity")
    g_sse = _tracker(capi, FirstLevel=3, LastLevel=0).match(pair["gr"], pair["gc"])
    g_exact = capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=0)).match(pair["gr"], pair["gc"])
    o_sse = orc.match(orc.default_config(first_level=3, last_level=0, rcp_mode=orc.RCP_SSE), pair["orr"], pair["occ"])
    lines.append(f"[rcpps] headline pair, the natively built oracle (-O3 -march=native, fma-contracting): vs the portable oracle "
                 f"(both rcpps) {synth.pose_error(o_sse['T'], n_sse['T']):.2e}; GPU(host rcpps) vs native(rcpps) "
                 f"{synth.pose_error(n_sse['T'], g_sse.Transformation):.2e}; GPU(exact) vs native(exact) "
                 f"{synth.pose_error(n_exact['T'], g_exact.Transformation):.2e}; GPU(exact) vs native(rcpps) "
                 f"{synth.pose_error(n_sse['T'], g_exact.Transformation):.2e}")
    with capsys.disabled():
        print()
        for ln in lines:
            print(ln)


def test_mode_is_per_tracker_deterministic_and_refused_while_queued(capi, synth, pair):
    a, b = _tracker(capi, FirstLevel=3, LastLevel=0), _tracker(capi, FirstLevel=3, LastLevel=0)
    ra = a.match(pair["gr"], pair["gc"])
    out = b.match_batch([pair["gr"]] * 20, [pair["gc"]] * 20, in_flight=12)
    assert all(np.array_equal(ra.Transformation, r.Transformation) for r in out)  # a function of the inputs in this mode too
    assert np.array_equal(a.match_banded(pair["gr"], pair["gc"], 4).Transformation, ra.Transformation)
    plain = capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=0)).match(pair["gr"], pair["gc"])
    assert not np.array_equal(plain.Transformation, ra.Transformation)
    assert synth.pose_error(plain.Transformation, ra.Transformation) < 3e-4
    sub = b.submit([pair["gr"]] * 30, [pair["gc"]] * 30, in_flight=10)
    with pytest.raises(capi.DvoAmdError):
        b.set_reciprocal_mode("exact")
    b.wait(sub)
    b.set_reciprocal_mode("exact")
    assert np.array_equal(b.match(pair["gr"], pair["gc"]).Transformation, plain.Transformation)
