"""ctypes binding of oracle/dvo_oracle.c -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libdvo_oracle.so")

MAX_LEVELS = 8
RCP_SSE, RCP_EXACT, RCP_CLEAN = 0, 1, 2
# orc_config.sum_mode: the order the same fp32 terms are added up in (test instrumentation)
SUM_REFERENCE, SUM_FP64, SUM_BLOCKED, SUM_BLOCKED_32, SUM_BLOCKED_2048 = 0, 1, 2, 3, 4
TERMINATION = {0: "IterationsExceeded", 1: "IncrementTooSmall", 2: "LogLikelihoodDecreased", 3: "TooFewConstraints", -1: "Unset"}


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("dvo_oracle.c", "dvo_oracle.h", "Makefile")]
    if force or not os.path.exists(_LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src):
        subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)
    return _LIB_PATH


def _host_tag() -> str:
    """-march=native code only runs on the CPU model it was built on: key the timing build by the host's model name"""
    import hashlib

    model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return hashlib.sha1(model.encode()).hexdigest()[:10]


def build_native() -> str:
    """The TIMING build of the same source with the reference's own flags, `-O3 -march=native` + SSE
    (dvo_core/CMakeLists.txt:35-40), compiled on the host that runs it.  Only bench.py's cpu_baseline legs use it; the
    parity checker is always the `build()` flavour (-mno-fma -ffp-contract=off -frounding-math: portable bits)."""
    path = os.path.join(_HERE, "_build", f"libdvo_oracle_native_{_host_tag()}.so")
    src = [os.path.join(_HERE, f) for f in ("dvo_oracle.c", "dvo_oracle.h")]
    if not os.path.exists(path) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in src):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        subprocess.run([os.environ.get("CC", "gcc"), "-O3", "-march=native", "-msse3", "-fPIC", "-std=gnu11", "-shared", "-o",
                        path + ".tmp", src[0], "-lm", "-lpthread"], check=True)
        os.replace(path + ".tmp", path)
    return path


def select_build(kind: str = "parity"):
    """Switch this module between the parity build (default) and the native timing build.  Objects created under one build
    must not be used under the other."""
    global _lib, _lib_override
    assert kind in ("parity", "native")
    _lib = None
    _lib_override = build_native() if kind == "native" else None


_lib_override = None


class Config(C.Structure):
    _fields_ = [("first_level", C.c_int), ("last_level", C.c_int), ("max_iterations_per_level", C.c_int),
                ("precision", C.c_double), ("mu", C.c_double), ("use_initial_estimate", C.c_int),
                ("intensity_derivative_threshold", C.c_float), ("depth_derivative_threshold", C.c_float),
                ("rcp_mode", C.c_int), ("sum_mode", C.c_int), ("ll_guard", C.c_int)]


class IterationStats(C.Structure):
    _fields_ = [("id", C.c_int), ("valid_constraints", C.c_int), ("tdist_loglik", C.c_double),
                ("tdist_mean", C.c_double * 2), ("tdist_precision", C.c_double * 4), ("prior_loglik", C.c_double),
                ("increment", C.c_double * 6), ("information", C.c_double * 36), ("rhs", C.c_double * 6),
                ("scale", C.c_float * 4), ("has_increment", C.c_int), ("estimate", C.c_double * 16),
                ("initial", C.c_double * 16)]


class MatchState(C.Structure):
    """orc_match_state (dvo_oracle.h): the state DenseTracker::match holds at the top of an iteration body"""
    _fields_ = [("level", C.c_int), ("iteration", C.c_int), ("estimate", C.c_double * 16), ("initial", C.c_double * 16),
                ("x", C.c_double * 6), ("last_error", C.c_double), ("precision", C.c_float * 4), ("has_previous", C.c_int),
                ("previous_information", C.c_double * 36), ("previous_loglik", C.c_double)]


class LevelStats(C.Structure):
    _fields_ = [("id", C.c_int), ("max_valid_pixels", C.c_int), ("valid_pixels", C.c_int), ("termination", C.c_int),
                ("n_iterations", C.c_int), ("first_iteration", C.c_int)]


class Result(C.Structure):
    _fields_ = [("T", C.c_double * 16), ("information", C.c_double * 36), ("loglik", C.c_double), ("n_levels", C.c_int),
                ("levels", LevelStats * MAX_LEVELS), ("n_iterations", C.c_int), ("iterations", C.POINTER(IterationStats)),
                ("iterations_capacity", C.c_int), ("is_nan", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(_lib_override or build())
        fp = C.POINTER(C.c_float)
        L.orc_default_config.argtypes = [C.POINTER(Config)]
        L.orc_pyramid_create.restype = C.c_void_p
        L.orc_pyramid_create.argtypes = [fp, fp, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int]
        L.orc_pyramid_destroy.argtypes = [C.c_void_p]
        L.orc_pyramid_levels.argtypes = [C.c_void_p]
        L.orc_level_size.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_level_intrinsics.argtypes = [C.c_void_p, C.c_int, fp]
        L.orc_level_plane.restype = fp
        L.orc_level_plane.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_select.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.POINTER(fp)]
        L.orc_select_index.restype = C.POINTER(C.c_int)
        L.orc_select_index.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float]
        L.orc_compute_residuals.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, fp, C.c_int, fp, fp,
                                            C.POINTER(C.c_ubyte)]
        L.orc_match.argtypes = [C.POINTER(Config), C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(Result)]
        L.orc_match_from.argtypes = [C.POINTER(Config), C.c_void_p, C.c_void_p, C.POINTER(MatchState), C.POINTER(Result)]
        L.orc_bench_threads.restype = C.c_longlong
        L.orc_bench_threads.argtypes = [C.POINTER(Config), C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_double,
                                        C.POINTER(C.c_double)]
        L.orc_ingest_depth_u16.argtypes = [C.POINTER(C.c_ushort), C.c_int, C.c_int, C.c_int, C.c_float, fp]
        L.orc_ingest_gray_from_bgr8.argtypes = [C.POINTER(C.c_ubyte), C.c_int, C.c_int, C.c_int, fp]
        L.orc_ingest_gray_from_gray8.argtypes = [C.POINTER(C.c_ubyte), C.c_int, C.c_int, C.c_int, fp]
        L.orc_se3_exp.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_se3_log.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_jacobian.argtypes = [fp, fp, fp]
        L.orc_rank_update.argtypes = [fp, fp, C.c_int, fp]
        L.orc_weights_scale_loglik.restype = C.c_float
        L.orc_weights_scale_loglik.argtypes = [fp, C.c_int, fp, C.c_int, C.c_int, fp, fp, fp]
        L.orc_iteration.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, fp, fp, C.c_int, C.c_int, fp, fp, fp, fp, fp]
        L.orc_host_rcp.restype = C.c_float
        L.orc_host_rcp.argtypes = [C.c_float]
        L.orc_host_rcp_many.restype = None
        L.orc_host_rcp_many.argtypes = [fp, fp, C.c_int]
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def default_config(**kw) -> Config:
    c = Config()
    lib().orc_default_config(C.byref(c))
    for k, v in kw.items():
        setattr(c, k, v)
    return c


class Pyramid:
    """RgbdImagePyramid with every level, derivative plane and point cloud built eagerly."""

    def __init__(self, intensity, depth, K, levels):
        intensity = np.ascontiguousarray(intensity, dtype=np.float32)
        depth = np.ascontiguousarray(depth, dtype=np.float32)
        assert intensity.shape == depth.shape and intensity.ndim == 2
        h, w = intensity.shape
        fx, fy, ox, oy = [float(k) for k in K]
        self.h = lib().orc_pyramid_create(_fp(intensity), _fp(depth), w, h, fx, fy, ox, oy, levels)
        if not self.h:
            raise ValueError("orc_pyramid_create failed")
        self.levels = levels

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_pyramid_destroy(self.h)
            self.h = None

    def size(self, level):
        w, h = C.c_int(), C.c_int()
        lib().orc_level_size(self.h, level, C.byref(w), C.byref(h))
        return w.value, h.value

    def intrinsics(self, level):
        k = np.zeros(4, np.float32)
        lib().orc_level_intrinsics(self.h, level, _fp(k))
        return k

    def plane(self, level, plane):
        w, h = self.size(level)
        p = lib().orc_level_plane(self.h, level, plane)
        return np.ctypeslib.as_array(p, shape=(h, w)).copy()

    def select(self, level, ti=0.0, td=0.0):
        rec = C.POINTER(C.c_float)()
        n = lib().orc_select(self.h, level, ti, td, C.byref(rec))
        idx = lib().orc_select_index(self.h, level, ti, td)
        if n == 0:
            return np.zeros((0, 12), np.float32), np.zeros(0, np.int32)
        return (np.ctypeslib.as_array(rec, shape=(n, 12)).copy(), np.ctypeslib.as_array(idx, shape=(n,)).copy())


def compute_residuals(ref: Pyramid, cur: Pyramid, level, T, rcp_mode=RCP_EXACT, ti=0.0, td=0.0):
    """Returns (points_error[n,12], residuals[n,2], valid[n_processed])."""
    n_sel = lib().orc_select(ref.h, level, ti, td, None)
    pe = np.zeros((max(n_sel, 1), 12), np.float32)
    r = np.zeros((max(n_sel, 1), 2), np.float32)
    valid = np.zeros(max(n_sel, 1), np.uint8)
    Tf = np.ascontiguousarray(np.asarray(T, dtype=np.float64).astype(np.float32).T)  # column-major
    n = lib().orc_compute_residuals(ref.h, cur.h, level, ti, td, _fp(Tf), rcp_mode, _fp(pe), _fp(r),
                                    valid.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return pe[:n].copy(), r[:n].copy(), valid[: n_sel - (n_sel % 2)].copy()


def iteration(ref: Pyramid, cur: Pyramid, level, T, prec_in=None, rcp_mode=RCP_EXACT, ti=0.0, td=0.0):
    """One Gauss-Newton iteration body at a fixed pose (dense_tracking.cpp:271-347 minus the accept test and the solve):
    unit weights when prec_in is None (first iteration of a level), else t-distribution weights from prec_in (2x2).
    Returns dict(n, scale 2x2, precision 2x2, ll, A 6x6, b 6)."""
    Tf = np.ascontiguousarray(np.asarray(T, dtype=np.float64).astype(np.float32).T)
    pin = np.zeros(4, np.float32) if prec_in is None else np.ascontiguousarray(np.asarray(prec_in, np.float32).T).ravel()
    scale, prec, A, b = np.zeros(4, np.float32), np.zeros(4, np.float32), np.zeros(36, np.float32), np.zeros(6, np.float32)
    ll = np.zeros(1, np.float32)
    n = lib().orc_iteration(ref.h, cur.h, level, ti, td, _fp(Tf), _fp(pin), int(prec_in is None), rcp_mode, _fp(scale),
                            _fp(prec), _fp(ll), _fp(A), _fp(b))
    return {"n": n, "scale": scale.reshape(2, 2).T.copy(), "precision": prec.reshape(2, 2).T.copy(), "ll": float(ll[0]),
            "A": A.reshape(6, 6).T.copy(), "b": b.copy()}


def match(cfg: Config, ref: Pyramid, cur: Pyramid, T_init=None):
    """DenseTracker::match.  Returns a dict with T (4x4), information (6x6), loglik, levels, iterations."""
    return _run_match(cfg, ref, cur, T_init, None)


def match_from(cfg: Config, ref: Pyramid, cur: Pyramid, *, level, iteration, estimate, initial, x, last_error=None, precision=None,
               previous_information=None, previous_loglik=None):
    """orc_match_from: the reference's control flow continued from the state it holds at the top of iteration `iteration` of
    `level` (test instrumentation, dvo_oracle.h).  estimate / initial: 4x4 before the increment x is applied; last_error None =
    DBL_MAX (a level start); precision: the previous iteration's 2x2.  Returns match()'s dict; levels[0] is `level` and holds only
    the iterations run here."""
    st = MatchState()
    st.level, st.iteration = int(level), int(iteration)
    st.estimate[:] = np.asarray(estimate, np.float64).T.ravel()
    st.initial[:] = np.asarray(initial, np.float64).T.ravel()
    st.x[:] = np.asarray(x, np.float64).ravel()
    st.last_error = float(np.finfo(np.float64).max) if last_error is None else float(last_error)
    st.precision[:] = (np.zeros(4, np.float32) if precision is None else np.asarray(precision, np.float32).T.ravel())
    st.has_previous = int(previous_information is not None)
    if previous_information is not None:
        st.previous_information[:] = np.asarray(previous_information, np.float64).T.ravel()
        st.previous_loglik = float(previous_loglik)
    return _run_match(cfg, ref, cur, None, st)


def _run_match(cfg, ref, cur, T_init, state):
    cap = (cfg.first_level - cfg.last_level + 1) * (cfg.max_iterations_per_level + 1)
    its = (IterationStats * cap)()
    res = Result()
    res.iterations = C.cast(its, C.POINTER(IterationStats))
    res.iterations_capacity = cap
    T0 = None
    if T_init is not None:
        T0a = np.ascontiguousarray(np.asarray(T_init, dtype=np.float64).T)
        T0 = T0a.ctypes.data_as(C.POINTER(C.c_double))
    if state is None:
        rc = lib().orc_match(C.byref(cfg), ref.h, cur.h, T0, C.byref(res))
    else:
        rc = lib().orc_match_from(C.byref(cfg), ref.h, cur.h, C.byref(state), C.byref(res))
    if rc != 0:
        raise RuntimeError(f"orc_match{'_from' if state is not None else ''} failed: {rc}")
    out = {
        "T": np.array(res.T[:]).reshape(4, 4).T.copy(),
        "information": np.array(res.information[:]).reshape(6, 6).T.copy(),
        "loglik": res.loglik,
        "is_nan": bool(res.is_nan),
        "levels": [],
    }
    for l in range(res.n_levels):
        L = res.levels[l]
        iters = []
        for k in range(L.n_iterations):
            it = its[L.first_iteration + k]
            iters.append({
                "id": it.id, "valid_constraints": it.valid_constraints, "tdist_loglik": it.tdist_loglik,
                "precision": np.array(it.tdist_precision[:]).reshape(2, 2).T.copy(), "prior_loglik": it.prior_loglik,
                "has_increment": bool(it.has_increment), "increment": np.array(it.increment[:]),
                "information": np.array(it.information[:]).reshape(6, 6).T.copy(), "rhs": np.array(it.rhs[:]),
                "scale": np.array(it.scale[:]).reshape(2, 2).T.copy(),
                "estimate": np.array(it.estimate[:]).reshape(4, 4).T.copy(),
                "initial": np.array(it.initial[:]).reshape(4, 4).T.copy(),
            })
        out["levels"].append({"id": L.id, "max_valid_pixels": L.max_valid_pixels, "valid_pixels": L.valid_pixels,
                              "termination": L.termination, "iterations": iters})
    return out


def bench_threads(cfg: Config, ref: Pyramid, curs, n_threads: int, seconds: float):
    """orc_bench_threads: (alignments finished by all threads, wall seconds); the threads run in C, no interpreter in the loop"""
    match(cfg, ref, curs[0])  # builds ref's selection before the threads share it
    arr = (C.c_void_p * len(curs))(*[c.h for c in curs])
    dt = C.c_double()
    n = lib().orc_bench_threads(C.byref(cfg), ref.h, arr, len(curs), int(n_threads), float(seconds), C.byref(dt))
    if n < 0:
        raise RuntimeError(f"orc_bench_threads failed: {n}")
    return int(n), dt.value


def se3_exp(xi):
    xi = np.ascontiguousarray(xi, dtype=np.float64)
    T = np.zeros(16)
    lib().orc_se3_exp(xi.ctypes.data_as(C.POINTER(C.c_double)), T.ctypes.data_as(C.POINTER(C.c_double)))
    return T.reshape(4, 4).T.copy()


def se3_log(T):
    Tc = np.ascontiguousarray(np.asarray(T, dtype=np.float64).T)
    xi = np.zeros(6)
    lib().orc_se3_log(Tc.ctypes.data_as(C.POINTER(C.c_double)), xi.ctypes.data_as(C.POINTER(C.c_double)))
    return xi


def ingest_depth(raw, scale=1.0 / 5000.0):
    """uint16 depth image -> float metres, 0 -> NaN (surface_pyramid.cpp:44-105)."""
    raw = np.ascontiguousarray(raw, dtype=np.uint16)
    h, w = raw.shape
    out = np.empty((h, w), np.float32)
    lib().orc_ingest_depth_u16(raw.ctypes.data_as(C.POINTER(C.c_ushort)), w, h, w, np.float32(scale), _fp(out))
    return out


def ingest_gray(image):
    """uint8 HxW (gray) or HxWx3 (BGR) -> float gray 0..255 (cv::cvtColor(BGR2GRAY) + convertTo(CV_32F))."""
    image = np.ascontiguousarray(image, dtype=np.uint8)
    h, w = image.shape[:2]
    out = np.empty((h, w), np.float32)
    ptr = image.ctypes.data_as(C.POINTER(C.c_ubyte))
    if image.ndim == 3:
        assert image.shape[2] == 3
        lib().orc_ingest_gray_from_bgr8(ptr, w, h, 3 * w, _fp(out))
    else:
        lib().orc_ingest_gray_from_gray8(ptr, w, h, w, _fp(out))
    return out


def host_rcp(x) -> np.ndarray:
    """_mm_rcp_ps of every element on THIS host (float32 in, float32 out)"""
    a = np.ascontiguousarray(x, dtype=np.float32).ravel()
    out = np.empty_like(a)
    lib().orc_host_rcp_many(_fp(a), _fp(out), a.size)
    return out.reshape(np.shape(x))
