"""A float64 numpy restatement of one Gauss-Newton iteration body, shared by the CPU and GPU stage tests."""
import numpy as np


def f64_iteration(orc, orr, occ, level, T, prec_in, P_eval, ti=0.0, td=0.0, rcp_mode=None):
    """A third, independent restatement of one iteration body (numpy, float64 sums) on the oracle's bit-exact residual
    records: what the sums would be without any accumulation error.  Q5 pairing and Q6 cut included.  Returns
    (n, cov 2x2, A 6x6 under P_eval, b 6 under P_eval, Cauchy-Schwarz scale of b, ll under P_eval)."""
    # ti, td: the point-selection thresholds; rcp_mode: the reciprocal the residual records are formed with (default: exact)
    pe, res, _ = orc.compute_residuals(orr, occ, level, T, orc.RCP_EXACT if rcp_mode is None else rcp_mode, ti, td)
    r = res.astype(np.float64)
    n = len(r)
    if prec_in is None:
        wg = np.ones(n)
    elif rcp_mode is not None and rcp_mode == orc.RCP_SSE:
        # the rcpps mode's weights are 7 * rcpps(5 + d) (dense_tracking_impl.cpp:700), off the exact quotient by up to 2^-12 in ONE
        # direction per table cell: "the same terms, summed exactly" must take the weights as that mode forms them
        import ctypes as C
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
        res32 = np.ascontiguousarray(res, np.float32)
        pin = np.ascontiguousarray(np.asarray(prec_in, np.float32).T).ravel()
        w32, cov, P = np.zeros(max(n, 1), np.float32), np.zeros(4, np.float32), np.zeros(4, np.float32)
        orc.lib().orc_weights_scale_loglik(fp(res32), n, fp(pin), 0, rcp_mode, fp(w32), fp(cov), fp(P))
        wg = w32[:n].astype(np.float64)
    else:
        d = np.einsum("ni,ij,nj->n", r, np.asarray(prec_in, np.float64), r)
        wg = 7.0 / (5.0 + d)
    n2 = n - (n % 2)
    S = np.einsum("n,ni,nj->ij", wg[0:n2:2] + wg[1:n2:2], r[0:n2:2], r[0:n2:2])  # Q5: residual 2j serves both halves of a pair
    if n % 2:
        S += wg[-1] * np.outer(r[-1], r[-1])
    cov = S / (n - 3)
    x, y, z = (pe[:, k].astype(np.float64) for k in range(3))
    iz, iz2 = 1.0 / z, 1.0 / (z * z)
    zero = 0.0 * z
    Jw0 = np.stack([iz, zero, -x * iz2, -x * iz2 * y, 1.0 + x * x * iz2, -y * iz], 1)   # dense_tracking.cpp:448-466
    Jw1 = np.stack([zero, iz, -y * iz2, -(1.0 + y * y * iz2), x * y * iz2, x * iz], 1)
    Jz = np.stack([zero, zero, 1.0 + zero, y, -x, zero], 1)                             # :468-476
    e = pe[:, 4:].astype(np.float64)
    J = np.stack([e[:, 2:3] * Jw0 + e[:, 3:4] * Jw1, e[:, 4:5] * Jw0 + e[:, 5:6] * Jw1 - Jz], 1)  # :333-339
    P = np.asarray(P_eval, np.float64)
    A = np.einsum("n,nai,ab,nbj->ij", wg, J, P, J)
    b = -np.einsum("n,nai,ab,nb->i", wg, J, P, r)
    cs = np.sqrt(np.diag(A) * np.einsum("n,na,ab,nb->", wg, r, P, r))  # |b_k| <= sqrt(A_kk * sum w r^T P r)
    m = 50 * (n // 50)  # Q6
    ll = 0.5 * n * np.log(np.float32(P[0, 0] * P[1, 1] - P[1, 0] * P[0, 1])) \
        - 3.5 * np.log1p(0.2 * np.einsum("ni,ij,nj->n", r[:m], P, r[:m])).sum()
    return n, cov, A, b, cs, ll
