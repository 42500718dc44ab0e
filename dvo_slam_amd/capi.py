"""ctypes binding of include/dvo_amd.h plus a thin Python mirror of the reference interface.

``DenseTracker`` / ``RgbdImagePyramid`` / ``Config`` / ``Result`` carry the same names, argument meaning and error
behaviour as dvo::DenseTracker, dvo::core::RgbdImagePyramid and their nested types
(dvo_core/include/dvo/dense_tracking.h:39-213, dvo_core/include/dvo/core/rgbd_image.h:242-262), so that the parity
tests read like calls into the reference.  All compute goes through the C ABI of libdvo_amd.so (HIP, gfx950): there is
no CPU path here, and loading fails loudly if the library is missing.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _build

MAX_LEVELS = 8
TERMINATION = {0: "IterationsExceeded", 1: "IncrementTooSmall", 2: "LogLikelihoodDecreased", 3: "TooFewConstraints",
               -1: "Unset"}

EXPORTS = [
    "dvo_amd_abi_version", "dvo_amd_build_id", "dvo_amd_status_string", "dvo_amd_last_error", "dvo_amd_device_count",
    "dvo_amd_default_config", "dvo_amd_context_create", "dvo_amd_context_destroy", "dvo_amd_configure",
    "dvo_amd_get_config", "dvo_amd_pyramid_create", "dvo_amd_pyramid_create_from_device", "dvo_amd_pyramid_retain",
    "dvo_amd_pyramid_release", "dvo_amd_pyramid_levels", "dvo_amd_pyramid_timestamp", "dvo_amd_pyramid_level_info",
    "dvo_amd_pyramid_download_plane", "dvo_amd_pyramid_select", "dvo_amd_match", "dvo_amd_match_batch",
    "dvo_amd_residuals", "dvo_amd_error_image", "dvo_amd_kernel_timing", "dvo_amd_se3_exp", "dvo_amd_se3_log",
    "dvo_amd_solve6", "dvo_amd_bench_residual_pass", "dvo_amd_match_many",
    "dvo_amd_debug_finalize_stamps", "dvo_amd_comm_unique_id", "dvo_amd_comm_create", "dvo_amd_comm_destroy",
    "dvo_amd_match_sharded", "dvo_amd_match_banded", "dvo_amd_context_device", "dvo_amd_debug_combine_bands", "dvo_amd_debug_wire_layout", "dvo_amd_debug_take_wire", "dvo_amd_pyramid_create_raw",
    "dvo_amd_default_validator_stages", "dvo_amd_proposals_for_candidates", "dvo_amd_validate_proposals",
    "dvo_amd_track_frame", "dvo_amd_png_info", "dvo_amd_png_read_bgr8", "dvo_amd_png_read_gray16",
    "dvo_amd_format_trajectory_line", "dvo_amd_debug_tick_log", "dvo_amd_debug_iteration", "dvo_amd_match_selection", "dvo_amd_bench_residual_pass_pairs",
    "dvo_amd_exchange_create", "dvo_amd_exchange_attach", "dvo_amd_exchange_destroy",
    "dvo_amd_match_submit", "dvo_amd_match_wait", "dvo_amd_match_poll", "dvo_amd_debug_next_seq",
    "dvo_amd_set_reciprocal_mode", "dvo_amd_get_reciprocal_mode", "dvo_amd_debug_rcp", "dvo_amd_debug_block_trace",
    "dvo_amd_debug_ll_overflow", "dvo_amd_debug_marker", "dvo_amd_debug_rcp_form", "dvo_amd_debug_weights", "dvo_amd_debug_hw_queue",
    "dvo_amd_debug_level_geometry",
]


class CConfig(C.Structure):
    _fields_ = [("first_level", C.c_int), ("last_level", C.c_int), ("max_iterations_per_level", C.c_int),
                ("precision", C.c_double), ("mu", C.c_double), ("use_initial_estimate", C.c_int),
                ("intensity_derivative_threshold", C.c_float), ("depth_derivative_threshold", C.c_float),
                ("segment_geometry", C.c_int), ("reserved", C.c_int)]


GEOMETRY_THROUGHPUT, GEOMETRY_LATENCY = 0, 1


class CIterationStats(C.Structure):
    _fields_ = [("id", C.c_int), ("valid_constraints", C.c_int), ("tdist_loglik", C.c_double),
                ("tdist_mean", C.c_double * 2), ("tdist_precision", C.c_double * 4), ("prior_loglik", C.c_double),
                ("increment", C.c_double * 6), ("information", C.c_double * 36), ("has_increment", C.c_int),
                ("reserved", C.c_int), ("estimate", C.c_double * 16), ("initial", C.c_double * 16)]


class CLevelStats(C.Structure):
    _fields_ = [("id", C.c_int), ("max_valid_pixels", C.c_int), ("valid_pixels", C.c_int), ("termination", C.c_int),
                ("n_iterations", C.c_int), ("first_iteration", C.c_int)]


class CResult(C.Structure):
    _fields_ = [("transformation", C.c_double * 16), ("information", C.c_double * 36), ("loglik", C.c_double),
                ("is_nan", C.c_int), ("n_levels", C.c_int), ("levels", CLevelStats * MAX_LEVELS),
                ("n_iterations", C.c_int), ("iterations_capacity", C.c_int),
                ("iterations", C.POINTER(CIterationStats)), ("n_ticks", C.c_int), ("n_residual_passes", C.c_int),
                ("alg_bytes", C.c_double), ("alg_bytes_discarded", C.c_double)]


class CFrameCriteria(C.Structure):
    _fields_ = [("odometry_is_nan", C.c_int), ("keyframe_is_nan", C.c_int), ("odometry_translation_norm", C.c_double),
                ("keyframe_translation_norm", C.c_double), ("keyframe_constraint_ratio", C.c_double),
                ("odometry_neg_loglik", C.c_double), ("keyframe_neg_loglik", C.c_double),
                ("odometry_condition_number", C.c_double), ("keyframe_condition_number", C.c_double)]


class CIterationProbe(C.Structure):
    _fields_ = [("valid_constraints", C.c_int), ("reserved", C.c_int), ("scale_sums", C.c_double * 3),
                ("scale", C.c_float * 4), ("precision", C.c_float * 4), ("moments", C.c_double * 87),
                ("information", C.c_double * 36), ("rhs", C.c_double * 6), ("loglik_sum", C.c_double),
                ("loglik", C.c_float), ("reserved_f", C.c_float)]


class CQ7Probe(C.Structure):
    _fields_ = [("n_tail", C.c_int), ("valid_constraints", C.c_int), ("valid_counted", C.c_int), ("recomputed_equal", C.c_int),
                ("pixel", C.c_int * 3), ("weight_table", C.c_float * 3), ("weight_exact", C.c_float * 3),
                ("scale_sums_delta", C.c_double * 3), ("moments_delta", C.c_double * 87)]


class DvoAmdError(RuntimeError):
    def __init__(self, status: int, where: str):
        L = lib()
        msg = L.dvo_amd_status_string(status).decode()
        detail = L.dvo_amd_last_error().decode()
        super().__init__(f"{where}: {msg}" + (f" [{detail}]" if detail else ""))
        self.status = status


_lib = None


def lib():
    """Load libdvo_amd.so (building it in-tree with hipcc if the sources are newer)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch wheels bundle their own ROCm runtime (libamdhip64 / libhsa-runtime64).  Two HIP runtimes in one process do not
    # share devices ("No HIP GPUs are available" in whichever initialises second), so when torch is installed let it load
    # its copy first: this library's DT_NEEDED libamdhip64.so.N then binds to that same copy.
    if os.environ.get("DVO_AMD_PRELOAD_TORCH", "1") != "0":
        try:
            import torch  # noqa: F401
        except Exception:  # torch absent or broken: the system runtime under /opt/rocm is used
            pass
    path = _build.LIB_PATH
    if _build.needs_build():
        # missing, or not built from exactly the sources next to it (dvo_amd_build_id() != the hash of csrc/* + flags)
        try:
            path = _build.build()
        except RuntimeError as exc:
            raise RuntimeError(f"{path} is missing or stale (build id {_build.library_id()!r}, sources {_build.source_id()!r}) "
                               f"and cannot be rebuilt here: {exc}") from exc
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: build it with `python -m dvo_slam_amd._build` (no CPU fallback exists)")
    L = C.CDLL(path)
    L.dvo_amd_build_id.restype = C.c_char_p
    if not os.environ.get("DVO_AMD_LIB") and L.dvo_amd_build_id().decode() != _build.source_id():
        raise RuntimeError(f"{path} carries build id {L.dvo_amd_build_id().decode()!r} but the sources next to it hash to "
                           f"{_build.source_id()!r}: refusing to run an edited tree against a stale binary")
    fp = C.POINTER(C.c_float)
    dp = C.POINTER(C.c_double)
    vp = C.c_void_p
    L.dvo_amd_abi_version.restype = C.c_int
    L.dvo_amd_status_string.restype = C.c_char_p
    L.dvo_amd_status_string.argtypes = [C.c_int]
    L.dvo_amd_last_error.restype = C.c_char_p
    L.dvo_amd_device_count.restype = C.c_int
    L.dvo_amd_default_config.argtypes = [C.POINTER(CConfig)]
    L.dvo_amd_context_create.argtypes = [C.c_int, C.POINTER(CConfig), C.POINTER(vp)]
    L.dvo_amd_context_destroy.argtypes = [vp]
    L.dvo_amd_context_destroy.restype = None
    L.dvo_amd_configure.argtypes = [vp, C.POINTER(CConfig)]
    L.dvo_amd_get_config.argtypes = [vp, C.POINTER(CConfig)]
    L.dvo_amd_set_reciprocal_mode.argtypes = [vp, C.c_int]
    L.dvo_amd_get_reciprocal_mode.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.dvo_amd_debug_rcp.argtypes = [vp, C.c_int, fp, fp]
    L.dvo_amd_debug_marker.argtypes = [vp, C.c_uint]
    L.dvo_amd_debug_rcp_form.argtypes = [vp, C.POINTER(C.c_int), C.c_char_p, C.c_int]
    L.dvo_amd_debug_ll_overflow.argtypes = [vp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp,
                                            C.POINTER(C.c_int)]
    L.dvo_amd_pyramid_create.argtypes = [C.c_int, fp, fp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                         C.c_float, C.c_int, C.c_double, C.POINTER(vp)]
    L.dvo_amd_pyramid_create_from_device.argtypes = [C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                                     C.c_float, C.c_float, C.c_int, C.c_double, C.POINTER(vp)]
    L.dvo_amd_pyramid_create_raw.argtypes = [C.c_int, vp, C.c_int, C.c_int, vp, C.c_int, C.c_float, C.c_int, C.c_int,
                                             C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_double,
                                             C.POINTER(vp)]
    L.dvo_amd_track_frame.argtypes = [vp, vp, vp, vp, dp, C.POINTER(CResult), C.POINTER(CResult), C.POINTER(CFrameCriteria)]
    L.dvo_amd_pyramid_retain.argtypes = [vp]
    L.dvo_amd_pyramid_retain.restype = None
    L.dvo_amd_pyramid_release.argtypes = [vp]
    L.dvo_amd_pyramid_release.restype = None
    L.dvo_amd_pyramid_levels.argtypes = [vp]
    L.dvo_amd_pyramid_timestamp.argtypes = [vp]
    L.dvo_amd_pyramid_timestamp.restype = C.c_double
    L.dvo_amd_pyramid_level_info.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), fp]
    L.dvo_amd_pyramid_download_plane.argtypes = [vp, C.c_int, C.c_int, fp]
    L.dvo_amd_pyramid_select.argtypes = [vp, C.c_int, C.c_float, C.c_float, C.POINTER(C.c_int), C.POINTER(C.c_ubyte)]
    L.dvo_amd_match.argtypes = [vp, vp, vp, dp, C.POINTER(CResult)]
    L.dvo_amd_match_selection.argtypes = [vp, vp, C.c_float, C.c_float, vp, dp, C.POINTER(CResult)]
    L.dvo_amd_match_batch.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(vp), dp, C.POINTER(CResult)]
    L.dvo_amd_match_many.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(vp), dp, C.POINTER(CResult), C.c_int]
    L.dvo_amd_match_submit.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(vp), dp, C.POINTER(CResult), C.c_int,
                                       C.POINTER(C.c_ulonglong)]
    L.dvo_amd_match_wait.argtypes = [vp, C.c_ulonglong]
    L.dvo_amd_match_poll.argtypes = [vp, C.c_ulonglong, C.POINTER(C.c_int)]
    L.dvo_amd_residuals.argtypes = [vp, vp, vp, C.c_int, fp, fp, C.POINTER(C.c_int)]
    L.dvo_amd_error_image.argtypes = [vp, vp, vp, dp, C.c_int, fp]
    L.dvo_amd_debug_iteration.argtypes = [vp, vp, vp, C.c_int, fp, fp, fp, C.POINTER(CIterationProbe)]
    L.dvo_amd_debug_hw_queue.argtypes = [vp, C.POINTER(C.c_int)]
    L.dvo_amd_debug_level_geometry.argtypes = [vp, vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.dvo_amd_debug_weights.argtypes = [vp, vp, vp, C.c_int, fp, fp, fp, C.POINTER(CQ7Probe)]
    L.dvo_amd_kernel_timing.argtypes = [vp, C.c_int, dp, C.POINTER(C.c_longlong), C.c_int]
    L.dvo_amd_bench_residual_pass.argtypes = [vp, vp, vp, C.c_int, fp, C.c_int, C.c_int, C.c_int, dp, dp,
                                              C.POINTER(C.c_int)]
    L.dvo_amd_bench_residual_pass_pairs.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(vp), C.c_int, fp, C.c_int, C.c_int, dp, dp,
                                                    C.POINTER(C.c_int)]
    L.dvo_amd_debug_finalize_stamps.argtypes = [vp, C.POINTER(C.c_ulonglong)]
    L.dvo_amd_comm_unique_id.argtypes = [C.POINTER(C.c_ubyte)]
    L.dvo_amd_comm_create.argtypes = [vp, C.POINTER(C.c_ubyte), C.c_int, C.c_int]
    L.dvo_amd_comm_destroy.argtypes = [vp]
    L.dvo_amd_comm_destroy.restype = None
    L.dvo_amd_exchange_create.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_ubyte)]
    L.dvo_amd_exchange_attach.argtypes = [vp, C.POINTER(C.c_ubyte)]
    L.dvo_amd_exchange_destroy.argtypes = [vp]
    L.dvo_amd_exchange_destroy.restype = None
    L.dvo_amd_match_sharded.argtypes = [vp, vp, vp, dp, C.POINTER(CResult)]
    L.dvo_amd_match_banded.argtypes = [vp, vp, vp, dp, C.POINTER(CResult), C.c_int]
    L.dvo_amd_debug_combine_bands.argtypes = [C.c_int, dp, dp]
    L.dvo_amd_debug_block_trace.restype = C.c_longlong
    L.dvo_amd_debug_block_trace.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_longlong]
    L.dvo_amd_debug_wire_layout.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.dvo_amd_debug_take_wire.argtypes = [C.POINTER(C.c_uint), C.c_uint, C.c_int, C.POINTER(C.c_uint)]
    L.dvo_amd_debug_next_seq.argtypes = [C.c_uint]
    L.dvo_amd_debug_next_seq.restype = C.c_uint
    L.dvo_amd_se3_exp.argtypes = [dp, dp]
    L.dvo_amd_se3_exp.restype = None
    L.dvo_amd_se3_log.argtypes = [dp, dp]
    L.dvo_amd_se3_log.restype = None
    L.dvo_amd_solve6.argtypes = [dp, dp, dp]
    L.dvo_amd_solve6.restype = None
    _lib = L
    return L


def build_id() -> str:
    """the hash of the sources and flags the loaded library was built from (dvo_amd_build_id)"""
    return lib().dvo_amd_build_id().decode()


def _check(status: int, where: str):
    if status != 0:
        raise DvoAmdError(status, where)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class Config:
    """DenseTracker::Config (live fields), defaults from dense_tracking_config.cpp:27-41."""

    def __init__(self, **kw):
        c = CConfig()
        lib().dvo_amd_default_config(C.byref(c))
        self.FirstLevel = c.first_level
        self.LastLevel = c.last_level
        self.MaxIterationsPerLevel = c.max_iterations_per_level
        self.Precision = c.precision
        self.Mu = c.mu
        self.UseInitialEstimate = bool(c.use_initial_estimate)
        self.IntensityDerivativeThreshold = c.intensity_derivative_threshold
        self.DepthDerivativeThreshold = c.depth_derivative_threshold
        # not a field of the reference: how a level is cut into wave segments (dvo_amd.h: dvo_amd_config::segment_geometry);
        # GEOMETRY_THROUGHPUT (default) or GEOMETRY_LATENCY (the shortest single match())
        self.SegmentGeometry = c.segment_geometry
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError(k)
            setattr(self, k, v)

    def getNumLevels(self) -> int:
        return self.FirstLevel + 1

    def IsSane(self) -> bool:
        return self.FirstLevel >= self.LastLevel

    def _c(self) -> CConfig:
        return CConfig(self.FirstLevel, self.LastLevel, self.MaxIterationsPerLevel, self.Precision, self.Mu,
                       int(self.UseInitialEstimate), self.IntensityDerivativeThreshold, self.DepthDerivativeThreshold,
                       int(self.SegmentGeometry), 0)


class RgbdImagePyramid:
    """RgbdCameraPyramid(w, h, K).create(intensity, depth) with `levels` levels built on the GPU."""

    def __init__(self, intensity, depth, K, levels: int, device: int = 0, timestamp: float = 0.0):
        intensity = np.ascontiguousarray(intensity, dtype=np.float32)
        depth = np.ascontiguousarray(depth, dtype=np.float32)
        if intensity.shape != depth.shape or intensity.ndim != 2:
            raise ValueError("intensity and depth must be 2-D arrays of the same shape")
        h, w = intensity.shape
        fx, fy, ox, oy = [float(k) for k in K]
        self._h = C.c_void_p()
        _check(lib().dvo_amd_pyramid_create(device, _fp(intensity), _fp(depth), w, h, w, fx, fy, ox, oy, levels,
                                            timestamp, C.byref(self._h)), "dvo_amd_pyramid_create")
        self.device = device

    @classmethod
    def from_device(cls, d_intensity: int, d_depth: int, width: int, height: int, K, levels: int, device: int = 0,
                    timestamp: float = 0.0, stride: int | None = None):
        """Planes already resident in HBM (raw device pointers, e.g. torch.Tensor.data_ptr())."""
        self = cls.__new__(cls)
        fx, fy, ox, oy = [float(k) for k in K]
        self._h = C.c_void_p()
        _check(lib().dvo_amd_pyramid_create_from_device(device, C.c_void_p(d_intensity), C.c_void_p(d_depth), width,
                                                        height, stride or width, fx, fy, ox, oy, levels, timestamp,
                                                        C.byref(self._h)), "dvo_amd_pyramid_create_from_device")
        self.device = device
        return self

    @classmethod
    def from_raw(cls, image, depth, K, levels: int, depth_scale: float = 1.0 / 5000.0, device: int = 0,
                 timestamp: float = 0.0):
        """Frame ingest on the device: uint8 image (HxW gray or HxWx3 BGR) + uint16 depth (0 = invalid), as a camera or a
        TUM PNG pair delivers them (benchmark_slam.cpp:46-93).  Gray conversion and depth scaling run on the GPU."""
        image = np.ascontiguousarray(image, dtype=np.uint8)
        depth = np.ascontiguousarray(depth, dtype=np.uint16)
        if image.shape[:2] != depth.shape or depth.ndim != 2 or image.ndim not in (2, 3):
            raise ValueError("image must be HxW or HxWx3 uint8 and depth HxW uint16 of the same size")
        channels = 1 if image.ndim == 2 else image.shape[2]
        h, w = depth.shape
        return cls._raw(image.ctypes.data, channels, w * channels, depth.ctypes.data, w, depth_scale, 0, w, h, K, levels,
                        device, timestamp)

    @classmethod
    def from_raw_device(cls, d_image: int, channels: int, d_depth: int, width: int, height: int, K, levels: int,
                        depth_scale: float = 1.0 / 5000.0, device: int = 0, timestamp: float = 0.0,
                        image_stride_bytes: int | None = None, depth_stride: int | None = None):
        """As from_raw, for raw frames already resident in HBM (device pointers)."""
        return cls._raw(d_image, channels, image_stride_bytes or width * channels, d_depth, depth_stride or width,
                        depth_scale, 1, width, height, K, levels, device, timestamp)

    @classmethod
    def _raw(cls, image_ptr, channels, image_stride, depth_ptr, depth_stride, depth_scale, on_device, w, h, K, levels,
             device, timestamp):
        self = cls.__new__(cls)
        fx, fy, ox, oy = [float(k) for k in K]
        self._h = C.c_void_p()
        _check(lib().dvo_amd_pyramid_create_raw(device, C.c_void_p(image_ptr), channels, image_stride,
                                                C.c_void_p(depth_ptr), depth_stride, depth_scale, on_device, w, h, fx, fy,
                                                ox, oy, levels, timestamp, C.byref(self._h)),
               "dvo_amd_pyramid_create_raw")
        self.device = device
        return self

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            lib().dvo_amd_pyramid_release(h)
            self._h = None

    def levels(self) -> int:
        return lib().dvo_amd_pyramid_levels(self._h)

    def timestamp(self) -> float:
        return lib().dvo_amd_pyramid_timestamp(self._h)

    def level_info(self, level: int):
        w, h = C.c_int(), C.c_int()
        k = np.zeros(4, np.float32)
        _check(lib().dvo_amd_pyramid_level_info(self._h, level, C.byref(w), C.byref(h), _fp(k)), "level_info")
        return w.value, h.value, k

    def plane(self, level: int, plane: int) -> np.ndarray:
        """0 I, 1 Z, 2 Ix, 3 Iy, 4 Zx, 5 Zy of a level, downloaded."""
        w, h, _ = self.level_info(level)
        out = np.empty((h, w), np.float32)
        _check(lib().dvo_amd_pyramid_download_plane(self._h, level, plane, _fp(out)), "download_plane")
        return out

    def select(self, level: int, ti: float = 0.0, td: float = 0.0):
        """PointSelection::select: (count, mask[h, w])."""
        w, h, _ = self.level_info(level)
        mask = np.empty((h, w), np.uint8)
        cnt = C.c_int()
        _check(lib().dvo_amd_pyramid_select(self._h, level, ti, td, C.byref(cnt),
                                            mask.ctypes.data_as(C.POINTER(C.c_ubyte))), "pyramid_select")
        return cnt.value, mask


class Result:
    """DenseTracker::Result: Transformation (4x4), Information (6x6), LogLikelihood, Statistics.Levels."""

    def __init__(self, c: CResult, its):
        self.Transformation = np.array(c.transformation[:]).reshape(4, 4).T.copy()
        self.Information = np.array(c.information[:]).reshape(6, 6).T.copy()
        self.LogLikelihood = c.loglik
        self._is_nan = bool(c.is_nan)
        self.n_ticks = c.n_ticks
        self.n_residual_passes = c.n_residual_passes
        self.alg_bytes = c.alg_bytes
        self.alg_bytes_discarded = c.alg_bytes_discarded
        self.Levels = []
        for l in range(c.n_levels):
            L = c.levels[l]
            iters = []
            for k in range(L.n_iterations if its is not None else 0):
                it = its[L.first_iteration + k]
                iters.append({
                    "Id": it.id, "ValidConstraints": it.valid_constraints,
                    "TDistributionLogLikelihood": it.tdist_loglik,
                    "TDistributionPrecision": np.array(it.tdist_precision[:]).reshape(2, 2).T.copy(),
                    "PriorLogLikelihood": it.prior_loglik, "has_increment": bool(it.has_increment),
                    "EstimateIncrement": np.array(it.increment[:]),
                    "EstimateInformation": np.array(it.information[:]).reshape(6, 6).T.copy(),
                    "estimate": np.array(it.estimate[:]).reshape(4, 4).T.copy(),  # instrumentation, not a reference field
                    "initial": np.array(it.initial[:]).reshape(4, 4).T.copy(),    # likewise
                })
            self.Levels.append({"Id": L.id, "MaxValidPixels": L.max_valid_pixels, "ValidPixels": L.valid_pixels,
                                "TerminationCriterion": L.termination, "Iterations": iters})

    def isNaN(self) -> bool:
        return self._is_nan


class Submission:
    """n pairs queued with DenseTracker.submit: the ticket, and the result structs the library fills (kept alive here)"""

    def __init__(self, ticket, res, its, n):
        self.ticket, self._res, self._its, self.n = ticket, res, its, n

    def results(self, raw: bool = False):
        if raw:
            self._res._keepalive = self._its
            return self._res
        return [Result(self._res[i], self._its[i]) for i in range(self.n)]


class DenseTracker:
    """dvo::DenseTracker: configure(), match(reference, current, T_init) -> Result.  One HIP stream; not thread-safe."""

    def __init__(self, config: Config | None = None, device: int = 0):
        self._cfg = config or Config()
        if not self._cfg.IsSane():
            raise DvoAmdError(5, "DenseTracker.configure")
        self._h = C.c_void_p()
        c = self._cfg._c()
        _check(lib().dvo_amd_context_create(device, C.byref(c), C.byref(self._h)), "dvo_amd_context_create")
        self.device = device
        # the library writes the results of a submission until it is complete: the tracker keeps every open Submission (its
        # result structs and iteration arrays) alive, whether or not the caller holds on to it
        self._open = {}

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            lib().dvo_amd_context_destroy(h)
            self._h = None

    def configuration(self) -> Config:
        return self._cfg

    def set_reciprocal_mode(self, mode: str):
        """"exact" (default): 1 / z of the projection is the exactly truncated quotient; "host_sse": the warp stage and the
        t-distribution weights use THIS HOST's _mm_rcp_ps bit for bit, from a table probed on the host (dvo_amd_set_reciprocal_mode)"""
        _check(lib().dvo_amd_set_reciprocal_mode(self._h, {"exact": 0, "host_sse": 1}[mode]), "dvo_amd_set_reciprocal_mode")

    def reciprocal_mode(self):
        """(mode name, mantissa bits the host's rcpps table is indexed by -- 0 in the exact mode)"""
        m, k = C.c_int(), C.c_int()
        _check(lib().dvo_amd_get_reciprocal_mode(self._h, C.byref(m), C.byref(k)), "dvo_amd_get_reciprocal_mode")
        return ("host_sse" if m.value else "exact"), k.value

    def reciprocal_form(self):
        """(diagnostic) ("off" | "table" | "nibbles", why the nibble form is not in use)"""
        form, note = C.c_int(), C.create_string_buffer(256)
        _check(lib().dvo_amd_debug_rcp_form(self._h, C.byref(form), note, 256), "dvo_amd_debug_rcp_form")
        return ("off", "table", "nibbles")[form.value], note.value.decode()

    def table_rcp(self, x) -> np.ndarray:
        """the table reciprocal of every element as the kernels compute it (test entry; needs the host_sse mode)"""
        a = np.ascontiguousarray(x, dtype=np.float32).ravel()
        out = np.empty_like(a)
        _check(lib().dvo_amd_debug_rcp(self._h, a.size, _fp(a), _fp(out)), "dvo_amd_debug_rcp")
        return out.reshape(np.shape(x))

    def configure(self, config: Config):
        c = config._c()
        _check(lib().dvo_amd_configure(self._h, C.byref(c)), "dvo_amd_configure")
        self._cfg = config

    def _alloc_results(self, n):
        cap = (self._cfg.FirstLevel - self._cfg.LastLevel + 1) * (self._cfg.MaxIterationsPerLevel + 1)
        res = (CResult * n)()
        its = []
        for i in range(n):
            buf = (CIterationStats * cap)()
            its.append(buf)
            res[i].iterations = C.cast(buf, C.POINTER(CIterationStats))
            res[i].iterations_capacity = cap
        return res, its

    def match(self, reference: RgbdImagePyramid, current: RgbdImagePyramid, T_init=None) -> Result:
        res, its = self._alloc_results(1)
        T0 = None
        if T_init is not None:
            T0a = np.ascontiguousarray(np.asarray(T_init, dtype=np.float64).T)
            T0 = T0a.ctypes.data_as(C.POINTER(C.c_double))
        _check(lib().dvo_amd_match(self._h, reference._h, current._h, T0, C.byref(res[0])), "dvo_amd_match")
        return Result(res[0], its[0])

    def alloc_results(self, n):
        """(result structs, per-iteration statistics arrays) for n pairs, reusable across match_batch(results=...) calls"""
        return self._alloc_results(n)

    def match_batch(self, references, currents, T_inits=None, stats: bool = True, in_flight: int = 0, raw: bool = False,
                    results=None):
        """n independent match() calls on this tracker's GPU.  in_flight = 0: all advanced in lock step; otherwise at most
        in_flight pairs are resident and a finished pair hands its slot to the next one.  raw=True returns the array of C
        result structs as the library filled them (no per-pair Python objects: for throughput loops).  results: a pair from
        alloc_results(n) to fill (per-iteration statistics included) instead of allocating new ones."""
        n = len(references)
        assert len(currents) == n
        if results is not None:
            res, its = results
        elif stats:
            res, its = self._alloc_results(n)
        else:
            res, its = (CResult * n)(), [None] * n
        refs = (C.c_void_p * n)(*[r._h for r in references])
        curs = (C.c_void_p * n)(*[c._h for c in currents])
        T0 = None
        if T_inits is not None:
            T0a = np.ascontiguousarray(np.stack([np.asarray(T, dtype=np.float64).T for T in T_inits]))
            T0 = T0a.ctypes.data_as(C.POINTER(C.c_double))
        _check(lib().dvo_amd_match_many(self._h, n, refs, curs, T0, res, in_flight), "dvo_amd_match_many")
        if raw:
            res._keepalive = its  # the iteration arrays the structs point into
            return res
        if not stats:
            return [Result(res[i], None) for i in range(n)]
        return [Result(res[i], its[i]) for i in range(n)]

    def submit(self, references, currents, T_inits=None, stats: bool = True, in_flight: int = 72, results=None):
        """dvo_amd_match_submit: queue n pairs behind whatever this tracker is still working on and return at once; the
        tracker keeps `in_flight` pairs resident across submissions (it never drains while it is fed).  Returns a
        Submission; wait(submission) / poll(submission) complete it."""
        n = len(references)
        assert len(currents) == n
        if results is not None:
            res, its = results
        elif stats:
            res, its = self._alloc_results(n)
        else:
            res, its = (CResult * n)(), [None] * n
        refs = (C.c_void_p * n)(*[r._h for r in references])
        curs = (C.c_void_p * n)(*[c._h for c in currents])
        T0, T0a = None, None
        if T_inits is not None:
            T0a = np.ascontiguousarray(np.stack([np.asarray(T, dtype=np.float64).T for T in T_inits]))
            T0 = T0a.ctypes.data_as(C.POINTER(C.c_double))
        ticket = C.c_ulonglong()
        _check(lib().dvo_amd_match_submit(self._h, n, refs, curs, T0, res, in_flight, C.byref(ticket)), "dvo_amd_match_submit")
        sub = Submission(ticket.value, res, its, n)
        self._open[sub.ticket] = sub
        return sub

    def wait(self, submission=None, raw: bool = False):
        """dvo_amd_match_wait: drive the queue until the submission (None: everything submitted) is complete; returns its
        results (raw=True: the C result structs as the library filled them)."""
        try:
            _check(lib().dvo_amd_match_wait(self._h, 0 if submission is None else submission.ticket), "dvo_amd_match_wait")
        finally:  # complete, or dropped by a failed tick: either way the library is done with the result storage
            if submission is None:
                self._open.clear()
            else:
                self._open.pop(submission.ticket, None)
        return None if submission is None else submission.results(raw)

    def poll(self, submission=None) -> bool:
        """dvo_amd_match_poll: advance whatever has landed, never waiting for the GPU; True when the submission is complete"""
        done = C.c_int()
        try:
            _check(lib().dvo_amd_match_poll(self._h, 0 if submission is None else submission.ticket, C.byref(done)),
                   "dvo_amd_match_poll")
        except DvoAmdError:
            self._open.clear()  # a failed tick drops everything queued
            raise
        if done.value:
            if submission is None:
                self._open.clear()
            else:
                self._open.pop(submission.ticket, None)
        return bool(done.value)

    def track_frame(self, keyframe: RgbdImagePyramid, last_frame: RgbdImagePyramid, frame: RgbdImagePyramid,
                    last_keyframe_pose=None):
        """The two alignments of LocalTracker::update (local_tracker.cpp:170-186) as one two-pair batch:
        (r_keyframe, r_odometry, criteria) with criteria = the inputs of KeyframeTracker's accept callbacks."""
        res, its = self._alloc_results(2)
        crit = CFrameCriteria()
        P = None
        if last_keyframe_pose is not None:
            P = np.ascontiguousarray(np.asarray(last_keyframe_pose, dtype=np.float64).T)
        _check(lib().dvo_amd_track_frame(self._h, keyframe._h, last_frame._h, frame._h,
                                         P.ctypes.data_as(C.POINTER(C.c_double)) if P is not None else None,
                                         C.byref(res[0]), C.byref(res[1]), C.byref(crit)), "dvo_amd_track_frame")
        criteria = {name: getattr(crit, name) for name, _ in CFrameCriteria._fields_}
        criteria["odometry_is_nan"], criteria["keyframe_is_nan"] = bool(crit.odometry_is_nan), bool(crit.keyframe_is_nan)
        return Result(res[0], its[0]), Result(res[1], its[1]), criteria

    def match_banded(self, reference: RgbdImagePyramid, current: RgbdImagePyramid, n_bands: int, T_init=None) -> Result:
        """The tile-shard pipeline with all n_bands bands on this one GPU (verification of the multi-GPU path)."""
        res, its = self._alloc_results(1)
        T0 = None
        if T_init is not None:
            T0a = np.ascontiguousarray(np.asarray(T_init, dtype=np.float64).T)
            T0 = T0a.ctypes.data_as(C.POINTER(C.c_double))
        _check(lib().dvo_amd_match_banded(self._h, reference._h, current._h, T0, C.byref(res[0]), n_bands), "dvo_amd_match_banded")
        return Result(res[0], its[0])

    def comm_create(self, unique_id: bytes, nranks: int, rank: int):
        """Attach an RCCL communicator (one rank per GPU) for match_sharded."""
        buf = (C.c_ubyte * 128).from_buffer_copy(unique_id)
        _check(lib().dvo_amd_comm_create(self._h, buf, nranks, rank), "dvo_amd_comm_create")

    def exchange_create(self, nranks: int, rank: int) -> bytes:
        """Allocate this rank's exchange buffer of the one-hop peer exchange; returns its 64-byte IPC handle."""
        buf = (C.c_ubyte * 64)()
        _check(lib().dvo_amd_exchange_create(self._h, nranks, rank, buf), "dvo_amd_exchange_create")
        return bytes(buf)

    def exchange_attach(self, handles):
        """handles: the 64-byte handles of all ranks in rank order (as all-gathered by the caller)."""
        blob = b"".join(handles)
        buf = (C.c_ubyte * len(blob)).from_buffer_copy(blob)
        _check(lib().dvo_amd_exchange_attach(self._h, buf), "dvo_amd_exchange_attach")

    def match_sharded(self, reference: RgbdImagePyramid, current: RgbdImagePyramid, T_init=None) -> Result:
        """One pair tile-sharded over the communicator's ranks; every rank calls this with the same arguments."""
        res, its = self._alloc_results(1)
        T0 = None
        if T_init is not None:
            T0a = np.ascontiguousarray(np.asarray(T_init, dtype=np.float64).T)
            T0 = T0a.ctypes.data_as(C.POINTER(C.c_double))
        _check(lib().dvo_amd_match_sharded(self._h, reference._h, current._h, T0, C.byref(res[0])), "dvo_amd_match_sharded")
        return Result(res[0], its[0])

    def residuals(self, reference: RgbdImagePyramid, current: RgbdImagePyramid, level: int, T):
        """computeResidualsAndValidFlagsSse: (residuals[h, w, 2] with NaN = invalid, n_valid)."""
        w, h, _ = reference.level_info(level)
        out = np.empty((h, w, 2), np.float32)
        Tf = np.ascontiguousarray(np.asarray(T, dtype=np.float64).astype(np.float32).T)
        n = C.c_int()
        _check(lib().dvo_amd_residuals(self._h, reference._h, current._h, level, _fp(Tf), _fp(out), C.byref(n)),
               "dvo_amd_residuals")
        return out, n.value

    def iteration_probe(self, reference: RgbdImagePyramid, current: RgbdImagePyramid, level: int, T, precision_in=None,
                        precision_eval=None):
        """One Gauss-Newton iteration body at the fixed pose T (dense_tracking.cpp:271-347 minus accept test and solve)
        through the kernels and host arithmetic match() uses: unit weights when precision_in is None, else t-distribution
        weights from the 2x2 precision_in.  precision_eval: evaluate A / b / ll with this 2x2 precision instead of the computed
        one.  Returns dict(n, scale, precision, ll, A, b, moments, scale_sums)."""
        Tf = np.ascontiguousarray(np.asarray(T, dtype=np.float64).astype(np.float32).T)
        pin = None if precision_in is None else np.ascontiguousarray(np.asarray(precision_in, np.float32).T).ravel()
        pev = None if precision_eval is None else np.ascontiguousarray(np.asarray(precision_eval, np.float32).T).ravel()
        pr = CIterationProbe()
        _check(lib().dvo_amd_debug_iteration(self._h, reference._h, current._h, level, _fp(Tf),
                                             None if pin is None else _fp(pin), None if pev is None else _fp(pev),
                                             C.byref(pr)), "dvo_amd_debug_iteration")
        return {"n": pr.valid_constraints, "scale": np.array(pr.scale[:], np.float32).reshape(2, 2).T.copy(),
                "precision": np.array(pr.precision[:], np.float32).reshape(2, 2).T.copy(), "ll": float(pr.loglik),
                "A": np.array(pr.information[:]).reshape(6, 6).T.copy(), "b": np.array(pr.rhs[:]),
                "moments": np.array(pr.moments[:]), "scale_sums": np.array(pr.scale_sums[:]), "ll_sum": pr.loglik_sum}

    def hw_queue(self) -> int:
        """(diagnostic, dvo_amd_debug.h) pipe << 3 | queue of the hardware queue this tracker's main stream runs on, asked of the GPU"""
        q = C.c_int(-2)
        _check(lib().dvo_amd_debug_hw_queue(self._h, C.byref(q)), "dvo_amd_debug_hw_queue")
        return q.value

    def level_geometry(self, reference: RgbdImagePyramid, level: int):
        """(test entry, dvo_amd_debug.h) (64-point steps per wave segment, blocks of four segments, points the pass walks) of the
        residual pass on one level of `reference` under this tracker's configuration"""
        steps, blocks, points = C.c_int(0), C.c_int(0), C.c_int(0)
        _check(lib().dvo_amd_debug_level_geometry(self._h, reference._h, level, C.byref(steps), C.byref(blocks), C.byref(points)),
               "dvo_amd_debug_level_geometry")
        return steps.value, blocks.value, points.value

    def weights_probe(self, reference: RgbdImagePyramid, current: RgbdImagePyramid, level: int, T, precision_in):
        """(test entry, dvo_amd_debug.h; host-rcpps mode only) the t-distribution weights of one residual pass at T under the 2x2
        precision_in as the product kernel formed them -- [h, w], NaN where the pixel is no constraint -- and what k_q7_tail did
        about the last V mod 4 of them (Q7): dict(n_tail, n, n_counted, recomputed_equal, pixel, w_table, w_exact, scale_sums_delta,
        moments_delta)"""
        w, h, _ = reference.level_info(level)
        out = np.empty((h, w), np.float32)
        Tf = np.ascontiguousarray(np.asarray(T, dtype=np.float64).astype(np.float32).T)
        pin = np.ascontiguousarray(np.asarray(precision_in, np.float32).T).ravel()
        q = CQ7Probe()
        _check(lib().dvo_amd_debug_weights(self._h, reference._h, current._h, level, _fp(Tf), _fp(pin), _fp(out), C.byref(q)),
               "dvo_amd_debug_weights")
        return out, {"n_tail": q.n_tail, "n": q.valid_constraints, "n_counted": q.valid_counted,
                     "recomputed_equal": bool(q.recomputed_equal), "pixel": list(q.pixel[:]),
                     "w_table": np.array(q.weight_table[:], np.float32), "w_exact": np.array(q.weight_exact[:], np.float32),
                     "scale_sums_delta": np.array(q.scale_sums_delta[:]), "moments_delta": np.array(q.moments_delta[:])}

    def ll_overflow_probe(self, residuals, n_blocks: int, steps: int, seg_first: int, n_segs: int, rank_offset: int, rank_end: int,
                          cut_rank: int, precision) -> bool:
        """(test entry, dvo_amd_debug.h) k_ll_overflow over a caller-supplied residual buffer [n_blocks * 4 * 64 * steps, 2] for
        the band of wave segments [seg_first, seg_first + n_segs); rank_end >= 0: a closed band"""
        r = np.ascontiguousarray(residuals, dtype=np.float32)
        assert r.shape == (n_blocks * 4 * 64 * steps, 2)
        P = np.ascontiguousarray(np.asarray(precision, np.float32).T).ravel()
        out = C.c_int(0)
        _check(lib().dvo_amd_debug_ll_overflow(self._h, _fp(r), n_blocks, steps, seg_first, n_segs, rank_offset, rank_end, cut_rank,
                                               _fp(P), C.byref(out)), "dvo_amd_debug_ll_overflow")
        return bool(out.value)

    def computeIntensityErrorImage(self, reference, current, T, level: int = 0) -> np.ndarray:
        w, h, _ = reference.level_info(level)
        out = np.empty((h, w), np.float32)
        Td = np.ascontiguousarray(np.asarray(T, dtype=np.float64).T)
        _check(lib().dvo_amd_error_image(self._h, reference._h, current._h, Td.ctypes.data_as(C.POINTER(C.c_double)),
                                         level, _fp(out)), "dvo_amd_error_image")
        return out

    def bench_residual_pass(self, reference, current, level: int, T, n_items: int, rounds: int = 0, reps: int = 20):
        """Time the fused residual-pass kernel alone.  Returns (avg ms per repetition, algorithmic bytes, launches)."""
        Tf = np.ascontiguousarray(np.asarray(T, dtype=np.float64).astype(np.float32).T)
        ms, ab, nl = C.c_double(), C.c_double(), C.c_int()
        _check(lib().dvo_amd_bench_residual_pass(self._h, reference._h, current._h, level, _fp(Tf), n_items, rounds, reps,
                                                 C.byref(ms), C.byref(ab), C.byref(nl)), "dvo_amd_bench_residual_pass")
        return ms.value, ab.value, nl.value

    def bench_residual_pass_pairs(self, references, currents, level: int, T, rounds: int = 0, reps: int = 20):
        """The same over different (reference, current) pairs: nothing for the caches to deduplicate."""
        n = len(references)
        Tf = np.ascontiguousarray(np.asarray(T, dtype=np.float64).astype(np.float32).T)
        refs = (C.c_void_p * n)(*[r._h for r in references])
        curs = (C.c_void_p * n)(*[c._h for c in currents])
        ms, ab, nl = C.c_double(), C.c_double(), C.c_int()
        _check(lib().dvo_amd_bench_residual_pass_pairs(self._h, n, refs, curs, level, _fp(Tf), rounds, reps, C.byref(ms),
                                                       C.byref(ab), C.byref(nl)), "dvo_amd_bench_residual_pass_pairs")
        return ms.value, ab.value, nl.value

    def tick_log(self) -> np.ndarray:
        """Per-launch log of the timed k_tick launches: rows {ms, items, residual blocks, likelihood blocks, grid.x, px,
        64-pixel wave steps of the residual items, 64-pixel wave steps of the likelihood items}."""
        n = C.c_int()
        L = lib()
        L.dvo_amd_debug_tick_log.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int)]
        _check(L.dvo_amd_debug_tick_log(self._h, None, 0, C.byref(n)), "tick_log")
        out = np.zeros((max(n.value, 1), 8))
        _check(L.dvo_amd_debug_tick_log(self._h, out.ctypes.data_as(C.POINTER(C.c_double)), n.value, C.byref(n)), "tick_log")
        return out[: n.value]

    def marker(self, tag: int = 0):
        """(profiling aid) a no-op dispatch named k_marker on this tracker's main stream (dvo_amd_debug.h)"""
        _check(lib().dvo_amd_debug_marker(self._h, int(tag)), "dvo_amd_debug_marker")

    def kernel_timing(self, enable: bool, reset: bool = False):
        ms = C.c_double()
        n = C.c_longlong()
        _check(lib().dvo_amd_kernel_timing(self._h, int(enable), C.byref(ms), C.byref(n), int(reset)), "kernel_timing")
        return ms.value, n.value


def se3_exp(xi) -> np.ndarray:
    xi = np.ascontiguousarray(xi, dtype=np.float64)
    T = np.zeros(16)
    lib().dvo_amd_se3_exp(xi.ctypes.data_as(C.POINTER(C.c_double)), T.ctypes.data_as(C.POINTER(C.c_double)))
    return T.reshape(4, 4).T.copy()


def se3_log(T) -> np.ndarray:
    Tc = np.ascontiguousarray(np.asarray(T, dtype=np.float64).T)
    xi = np.zeros(6)
    lib().dvo_amd_se3_log(Tc.ctypes.data_as(C.POINTER(C.c_double)), xi.ctypes.data_as(C.POINTER(C.c_double)))
    return xi


def solve6(A, b) -> np.ndarray:
    Ac = np.ascontiguousarray(np.asarray(A, dtype=np.float64).T)
    bc = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros(6)
    dp = C.POINTER(C.c_double)
    lib().dvo_amd_solve6(Ac.ctypes.data_as(dp), bc.ctypes.data_as(dp), x.ctypes.data_as(dp))
    return x


def comm_unique_id() -> bytes:
    buf = (C.c_ubyte * 128)()
    _check(lib().dvo_amd_comm_unique_id(buf), "dvo_amd_comm_unique_id")
    return bytes(buf)


def wire_layout():
    """(pieces, payload words) of a record on its way to the host: pieces of two {payload word, tick number} halves"""
    a, b = C.c_int(), C.c_int()
    _check(lib().dvo_amd_debug_wire_layout(C.byref(a), C.byref(b)), "wire_layout")
    return a.value, b.value


def take_wire(wire: np.ndarray, tick: int, from_piece: int, record: np.ndarray) -> int:
    """Host side of the record hand-off (host only): `wire` is a 16-byte aligned uint32 array [pieces, 4], `record` the uint32
    payload words collected so far; returns the first piece that does not carry `tick` yet."""
    up = C.POINTER(C.c_uint)
    assert wire.dtype == np.uint32 and record.dtype == np.uint32 and wire.flags.c_contiguous and record.flags.c_contiguous
    rc = lib().dvo_amd_debug_take_wire(wire.ctypes.data_as(up), tick, from_piece, record.ctypes.data_as(up))
    if rc < 0:
        _check(-rc, "take_wire")
    return rc


def combine_bands(bands) -> np.ndarray:
    """bands: [n, 10] = {valid, first_w, last_r0, last_r1, S[3], S_odd[3]} -> {valid, S[3], S_odd[3]} (host only)."""
    b = np.ascontiguousarray(bands, dtype=np.float64)
    out = np.zeros(7)
    dp = C.POINTER(C.c_double)
    _check(lib().dvo_amd_debug_combine_bands(b.shape[0], b.ctypes.data_as(dp), out.ctypes.data_as(dp)), "combine_bands")
    return out
