"""TUM RGB-D benchmark on-disk formats over the C ABI.

Mirrors what dvo_benchmark reads and writes around the tracker: the association file (`rgb_ts rgb_file depth_ts depth_file`
per line, rgbd_pair.h:59-71 read through file_reader.h:36-113), the frame loader `load()` of benchmark_slam.cpp:46-93
(cv::imread + gray conversion + depth scaling, here: PNG decode on the host, conversion on the GPU), and the estimated
trajectory (`timestamp tx ty tz qx qy qz qw`, benchmark_slam.cpp:490-504).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import capi

TUM_DEPTH_SCALE = 1.0 / 5000.0  # benchmark_slam.cpp:77
TUM_FR1_INTRINSICS = (517.3, 516.5, 318.6, 255.3)  # benchmark_slam.cpp:384

_bound = False


def _lib():
    global _bound
    L = capi.lib()
    if not _bound:
        ip = C.POINTER(C.c_int)
        L.dvo_amd_png_info.argtypes = [C.c_char_p, ip, ip, ip, ip]
        L.dvo_amd_png_read_bgr8.argtypes = [C.c_char_p, C.POINTER(C.c_ubyte), C.c_int, C.c_int]
        L.dvo_amd_png_read_gray16.argtypes = [C.c_char_p, C.POINTER(C.c_ushort), C.c_int, C.c_int]
        L.dvo_amd_format_trajectory_line.argtypes = [C.c_double, C.POINTER(C.c_double), C.c_char_p, C.c_int]
        _bound = True
    return L


def png_info(path: str):
    w, h, c, b = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    capi._check(_lib().dvo_amd_png_info(os.fsencode(path), C.byref(w), C.byref(h), C.byref(c), C.byref(b)), f"png_info({path})")
    return w.value, h.value, c.value, b.value


def imread_color(path: str) -> np.ndarray:
    """cv::imread(path, 1): HxWx3 uint8, B G R."""
    w, h, _, _ = png_info(path)
    out = np.empty((h, w, 3), np.uint8)
    capi._check(_lib().dvo_amd_png_read_bgr8(os.fsencode(path), out.ctypes.data_as(C.POINTER(C.c_ubyte)), w, h),
                f"png_read_bgr8({path})")
    return out


def imread_depth(path: str) -> np.ndarray:
    """cv::imread(path, -1) of a TUM depth image: HxW uint16."""
    w, h, _, _ = png_info(path)
    out = np.empty((h, w), np.uint16)
    capi._check(_lib().dvo_amd_png_read_gray16(os.fsencode(path), out.ctypes.data_as(C.POINTER(C.c_ushort)), w, h),
                f"png_read_gray16({path})")
    return out


class RgbdPair:
    """dvo_benchmark::RgbdPair (rgbd_pair.h:32-47)."""

    def __init__(self, rgb_timestamp: float, rgb_file: str, depth_timestamp: float, depth_file: str):
        self.RgbTimestamp, self.RgbFile, self.DepthTimestamp, self.DepthFile = rgb_timestamp, rgb_file, depth_timestamp, depth_file

    def __repr__(self):
        return f"RgbdPair({self.RgbTimestamp!r}, {self.RgbFile!r}, {self.DepthTimestamp!r}, {self.DepthFile!r})"


def read_assoc(path: str, reference_trailing_entry: bool = False):
    """FileReader<RgbdPair>::skipComments() + readAllEntries() (file_reader.h:63-102, benchmark_slam.cpp:168-169,399).
    Leading '#' lines are skipped; entries are whitespace-separated tokens, four per entry (line breaks do not matter to
    operator>>).  reference_trailing_entry=True reproduces the reference's end-of-file behaviour for a file that ends in a
    newline: one more next() succeeds on a stream that then fails, which appends a copy of the last entry whose
    timestamps are 0 (value-initialised by a failed C++11 extraction)."""
    with open(path, "r") as f:
        text = f.read()
    pos = 0
    while pos < len(text) and text[pos] == "#":  # skipComments: only while the NEXT character is '#'
        nl = text.find("\n", pos)
        pos = len(text) if nl < 0 else nl + 1
    body = text[pos:]
    tokens = body.split()
    entries = []
    for i in range(0, len(tokens) - len(tokens) % 4, 4):
        entries.append(RgbdPair(float(tokens[i]), tokens[i + 1], float(tokens[i + 2]), tokens[i + 3]))
    if reference_trailing_entry and entries and body[-1:].isspace():
        last = entries[-1]
        entries.append(RgbdPair(0.0, last.RgbFile, 0.0, last.DepthFile))
    return entries


def read_groundtruth(path: str):
    """groundtruth.txt / a trajectory file: [(timestamp, 4x4 pose)] from `timestamp tx ty tz qx qy qz qw` lines."""
    out = []
    with open(path, "r") as f:
        for line in f:
            t = line.split()
            if len(t) < 8 or line.lstrip().startswith("#"):
                continue
            ts, tx, ty, tz, qx, qy, qz, qw = [float(v) for v in t[:8]]
            n = np.sqrt(qx * qx + qy * qy + qz * qz + qw * qw)
            qx, qy, qz, qw = qx / n, qy / n, qz / n, qw / n
            T = np.eye(4)
            T[:3, :3] = [[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)],
                         [2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)],
                         [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)]]
            T[:3, 3] = [tx, ty, tz]
            out.append((ts, T))
    return out


def load(K, rgb_file: str, depth_file: str, levels: int, device: int = 0, timestamp: float = 0.0,
         depth_scale: float = TUM_DEPTH_SCALE) -> capi.RgbdImagePyramid:
    """`load()` of benchmark_slam.cpp:46-93: imread both files, BGR -> gray float, raw depth -> metres with 0 -> NaN,
    camera.create(...).  Decoding happens on the host, both conversions and the pyramid on the GPU."""
    bgr = imread_color(rgb_file)
    depth = imread_depth(depth_file)
    return capi.RgbdImagePyramid.from_raw(bgr, depth, K, levels, depth_scale=depth_scale, device=device, timestamp=timestamp)


def format_trajectory_line(timestamp: float, T) -> str:
    """One line of the estimated trajectory as benchmark_slam.cpp:490-504 prints it (including the trailing blank)."""
    Tc = np.ascontiguousarray(np.asarray(T, dtype=np.float64).T)
    buf = C.create_string_buffer(256)
    n = _lib().dvo_amd_format_trajectory_line(float(timestamp), Tc.ctypes.data_as(C.POINTER(C.c_double)), buf, 256)
    if n < 0:
        raise RuntimeError("dvo_amd_format_trajectory_line failed")
    return buf.raw[:n].decode()


def replay(assoc_path: str, trajectory_path: str | None = None, K=TUM_FR1_INTRINSICS, config: capi.Config | None = None,
           device: int = 0, max_frames: int | None = None):
    """Frame-to-frame odometry over an association file (the EstimateTrajectory mode of dvo_benchmark without the SLAM back
    end): pose_t = pose_{t-1} * T_t^-1 with T_t = match(frame_{t-1}, frame_t), one trajectory line per frame.
    Returns [(timestamp, 4x4 pose)]."""
    cfg = config or capi.Config(FirstLevel=3, LastLevel=1)
    trk = capi.DenseTracker(cfg, device=device)
    base = os.path.dirname(os.path.abspath(assoc_path))
    pairs = read_assoc(assoc_path)
    if max_frames is not None:
        pairs = pairs[:max_frames]
    pose = np.eye(4)
    prev = None
    out = []
    fh = open(trajectory_path, "w") if trajectory_path else None
    try:
        for p in pairs:
            cur = load(K, os.path.join(base, p.RgbFile), os.path.join(base, p.DepthFile), cfg.getNumLevels(), device,
                       p.RgbTimestamp)
            if prev is not None:
                r = trk.match(prev, cur)
                if not r.isNaN():
                    pose = pose @ np.linalg.inv(r.Transformation)
            out.append((p.RgbTimestamp, pose.copy()))
            if fh:
                fh.write(format_trajectory_line(p.RgbTimestamp, pose))
            prev = cur
    finally:
        if fh:
            fh.close()
    return out
