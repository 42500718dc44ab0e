"""Synthetic RGB-D frames for the dense-alignment hot path.

An analytic room corner (floor y=+1.2, wall z=+3.0, wall x=-1.6, all in the reference-camera
frame) is ray-cast from an arbitrary camera pose, so depth is exact and the ground-truth
relative pose of any two frames is known.  Texture = a smooth sinusoid + integer-hash value
noise of the in-plane coordinates; a hash-selected 2 % of the depth pixels and one 40x40 block
are NaN (invalid), as a structured-light sensor would produce.

Inputs follow the reference's contract (dvo_benchmark/src/benchmark_slam.cpp:46-93): intensity is
float32 in 0..255, depth is float32 metres with NaN = invalid, TUM fr1 intrinsics by default
(benchmark_slam.cpp:384).

Two regimes: render() / make_pair() are noise-free (exact depth); sensor_frame() / sensor_pair() deliver the same
views the way the reference's inputs arrive -- 8-bit grey, uint16 depth at 1/5000 m, depth noise of the sigma the
reference models itself (depth_std_dev) -- see the block above sensor_frame().
"""
from __future__ import annotations

import numpy as np

SEED = 20131103
TUM_FR1 = (517.3, 516.5, 318.6, 255.3)  # fx, fy, ox, oy at 640x480

XI_GT_PAIR = np.array([0.012, -0.006, 0.009, 0.004, -0.007, 0.003])      # config 2 / 3
XI_STEP_STREAM = np.array([0.004, -0.002, 0.003, 0.0015, -0.002, 0.001])  # config 4


def intrinsics_for(width: int, height: int):
    """TUM fr1 intrinsics scaled from 640x480 to (width, height)."""
    s = width / 640.0
    fx, fy, ox, oy = TUM_FR1
    return (np.float32(fx * s), np.float32(fy * s), np.float32(ox * s), np.float32(oy * s))


def _hat(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=np.float64)


def se3_exp(xi) -> np.ndarray:
    """4x4 matrix of exp(xi), xi = (upsilon, omega) -- the tangent order of Sophus::SE3d."""
    xi = np.asarray(xi, dtype=np.float64)
    ups, om = xi[:3], xi[3:]
    th = np.linalg.norm(om)
    O = _hat(om)
    if th < 1e-10:
        R = np.eye(3) + O + 0.5 * O @ O
        V = np.eye(3) + 0.5 * O + O @ O / 6.0
    else:
        R = np.eye(3) + np.sin(th) / th * O + (1 - np.cos(th)) / th**2 * O @ O
        V = np.eye(3) + (1 - np.cos(th)) / th**2 * O + (th - np.sin(th)) / th**3 * O @ O
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = V @ ups
    return T


def se3_log(T) -> np.ndarray:
    T = np.asarray(T, dtype=np.float64)
    R, t = T[:3, :3], T[:3, 3]
    c = np.clip((np.trace(R) - 1) / 2, -1, 1)
    th = np.arccos(c)
    if th < 1e-10:
        om = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / 2
    else:
        om = th / (2 * np.sin(th)) * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    O = _hat(om)
    if th < 1e-10:
        Vi = np.eye(3) - 0.5 * O + O @ O / 12.0
    else:
        Vi = np.eye(3) - 0.5 * O + (1 - th * np.cos(th / 2) / (2 * np.sin(th / 2))) / th**2 * O @ O
    return np.concatenate([Vi @ t, om])


def pose_error(T_a, T_b) -> float:
    """|| log(T_a^-1 T_b) ||_2 over the 6-vector (upsilon, omega): the parity metric of BASELINE.json."""
    return float(np.linalg.norm(se3_log(np.linalg.inv(np.asarray(T_a, dtype=np.float64)) @ np.asarray(T_b, dtype=np.float64))))


def _hash_u32(x, y, seed):
    h = (x.astype(np.uint32) * np.uint32(0x9E3779B1)) ^ (y.astype(np.uint32) * np.uint32(0x85EBCA77)) ^ np.uint32(seed & 0xFFFFFFFF)
    h ^= h >> np.uint32(15)
    h = h * np.uint32(0x2C1B3C6D)
    h ^= h >> np.uint32(12)
    h = h * np.uint32(0x297A2D39)
    h ^= h >> np.uint32(15)
    return h


def _value_noise(a, b, seed):
    a0 = np.floor(a)
    b0 = np.floor(b)
    fa = a - a0
    fb = b - b0
    ia = a0.astype(np.int64)
    ib = b0.astype(np.int64)

    def lattice(dx, dy):
        return _hash_u32((ia + dx) & 0xFFFFFFFF, (ib + dy) & 0xFFFFFFFF, seed).astype(np.float64) / 4294967296.0

    sa = fa * fa * (3 - 2 * fa)
    sb = fb * fb * (3 - 2 * fb)
    top = lattice(0, 0) * (1 - sa) + lattice(1, 0) * sa
    bot = lattice(0, 1) * (1 - sa) + lattice(1, 1) * sa
    return (top * (1 - sb) + bot * sb) * 2.0 - 1.0


def render(width: int, height: int, T_cam=None, seed: int = SEED, frame_id: int = 0, nan_fraction: float = 0.02,
           hole: bool = True, K=None):
    """Ray-cast the room corner from camera pose T_cam (camera -> scene, 4x4).  Returns (intensity, depth) float32."""
    fx, fy, ox, oy = (K if K is not None else intrinsics_for(width, height))
    T = np.eye(4) if T_cam is None else np.asarray(T_cam, dtype=np.float64)
    u, v = np.meshgrid(np.arange(width, dtype=np.float64), np.arange(height, dtype=np.float64))
    d_c = np.stack([(u - float(ox)) / float(fx), (v - float(oy)) / float(fy), np.ones_like(u)], axis=-1)
    d = d_c @ T[:3, :3].T
    o = T[:3, 3]
    big = 1e30
    with np.errstate(divide="ignore", invalid="ignore"):
        t_floor = np.where(d[..., 1] > 1e-9, (1.2 - o[1]) / d[..., 1], big)
        t_wz = np.where(d[..., 2] > 1e-9, (3.0 - o[2]) / d[..., 2], big)
        t_wx = np.where(d[..., 0] < -1e-9, (-1.6 - o[0]) / d[..., 0], big)
    t_floor = np.where(t_floor > 0, t_floor, big)
    t_wz = np.where(t_wz > 0, t_wz, big)
    t_wx = np.where(t_wx > 0, t_wx, big)
    t = np.minimum(np.minimum(t_floor, t_wz), t_wx)
    which = np.where(t == t_floor, 0, np.where(t == t_wz, 1, 2))
    P = o + d * t[..., None]
    a = np.where(which == 0, P[..., 0], np.where(which == 1, P[..., 0], P[..., 2]))
    b = np.where(which == 0, P[..., 2], P[..., 1])
    tex = 127.5 + 50.0 * np.sin(7.1 * a) * np.cos(5.3 * b)
    for pid in range(3):
        m = which == pid
        if m.any():
            tex[m] += 35.0 * _value_noise(a[m] * 8.0, b[m] * 8.0, seed + 17 * pid)
    intensity = np.clip(tex, 0.0, 255.0).astype(np.float32)
    depth = t.astype(np.float32)  # z in the camera frame because d_c.z == 1
    depth[t >= big] = np.nan
    if nan_fraction > 0:
        ui = u.astype(np.int64)
        vi = v.astype(np.int64)
        h = _hash_u32(ui, vi, seed + 1 + 7919 * frame_id)
        depth[h < np.uint32(int(nan_fraction * 4294967296.0))] = np.nan
    if hole and width >= 160 and height >= 120:
        s = max(1, width // 16)
        y0, x0 = height // 3, (2 * width) // 3
        depth[y0:y0 + s, x0:x0 + s] = np.nan
    return intensity, depth


def to_raw(intensity, depth, depth_factor: float = 5000.0, seed: int = SEED + 3):
    """The frame as a sensor / a TUM PNG pair would deliver it: uint8 BGR (the gray value spread over three channels with
    hashed per-channel offsets, so the colour conversion is exercised) and uint16 depth in 1/depth_factor metres with
    0 = no measurement (benchmark_slam.cpp:46-93)."""
    h, w = intensity.shape
    yy, xx = np.mgrid[0:h, 0:w].astype(np.uint32)
    base = np.rint(intensity).astype(np.int32)
    bgr = np.empty((h, w, 3), np.uint8)
    for c in range(3):
        off = (_hash_u32(xx, yy, seed + c) % np.uint32(31)).astype(np.int32) - 15
        bgr[..., c] = np.clip(base + off, 0, 255).astype(np.uint8)
    z = np.where(np.isnan(depth), 0.0, depth) * depth_factor
    raw_z = np.clip(np.rint(z), 0, 65535).astype(np.uint16)
    return bgr, raw_z


# ---- the regime the reference actually runs in: what a structured-light RGB-D sensor / a TUM PNG pair delivers ------------
# dvo_benchmark/src/benchmark_slam.cpp:56-80 feeds 8-bit grey (cv::cvtColor + convertTo(CV_32F), :60-68) and uint16 depth in
# 1/5000 m with 0 = no measurement (:77) from a sensor whose depth noise the reference models itself: depthStdDevZ(z) =
# 0.0012 + 0.0019 (z - 0.4)^2 (dvo_core/src/dense_tracking_impl.cpp:122-128, the occlusion test's sigma).  The analytic frames
# above are noise-free: their depth precision comes out as 1e9, the reference's 50-term likelihood product overflows and one
# valid pixel more or less flips an iteration -- artefacts that never occur on sensor data.  sensor_frame() adds what the
# sensor adds, deterministically (hashed, seeded): Gaussian depth noise of exactly that sigma, quantisation to 1/5000 m,
# Gaussian intensity noise and rounding to 8 bits.
DEPTH_FACTOR = 5000.0


def depth_std_dev(z):
    """depthStdDevZ, dense_tracking_impl.cpp:122-128"""
    z = np.asarray(z, dtype=np.float64)
    return 0.0012 + 0.0019 * (z - 0.4) ** 2


def _hash_normal(width: int, height: int, seed: int):
    """one N(0, 1) sample per pixel from two hashed uniforms (Box-Muller): deterministic, independent of numpy's generators"""
    u, v = np.meshgrid(np.arange(width, dtype=np.int64), np.arange(height, dtype=np.int64))
    u1 = (_hash_u32(u, v, seed).astype(np.float64) + 1.0) / 4294967297.0  # (0, 1)
    u2 = _hash_u32(u, v, seed ^ 0x5BD1E995).astype(np.float64) / 4294967296.0
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def sensor_from_analytic(intensity, depth, seed: int = SEED, frame_id: int = 0, intensity_sigma: float = 1.5,
                         depth_noise: float = 1.0, channels: int = 1):
    """A noise-free frame (render()) as the sensor would deliver it: depth = true depth + depth_noise * depthStdDevZ(depth) *
    N(0, 1), rounded to the 0.2 mm grid (0 = no measurement); grey = texture + intensity_sigma * N(0, 1), rounded and clipped
    to 0..255.  Noise is hashed from (pixel, seed, frame_id)."""
    height, width = intensity.shape
    nz = _hash_normal(width, height, seed + 1009 + 7919 * frame_id)
    ni = _hash_normal(width, height, seed + 2003 + 104729 * frame_id)
    z = depth.astype(np.float64)
    with np.errstate(invalid="ignore"):
        z_noisy = z + depth_noise * depth_std_dev(z) * nz
        raw = np.rint(np.where(np.isnan(z_noisy), 0.0, z_noisy) * DEPTH_FACTOR)
    raw_z = np.clip(raw, 0, 65535).astype(np.uint16)
    gray = np.clip(np.rint(intensity.astype(np.float64) + intensity_sigma * ni), 0, 255).astype(np.uint8)
    if channels == 3:
        gray = np.repeat(gray[..., None], 3, axis=2)  # a grey scene seen by a colour camera: B = G = R
    return gray, raw_z


def sensor_frame(width: int, height: int, T_cam=None, seed: int = SEED, frame_id: int = 0, intensity_sigma: float = 1.5,
                 depth_noise: float = 1.0, channels: int = 1, K=None, nan_fraction: float = 0.02, hole: bool = True):
    """The analytic view from T_cam as a sensor would deliver it: (uint8 grey HxW -- or BGR HxWx3 with channels=3 --, uint16
    depth in 1/5000 m, 0 = no measurement); see sensor_from_analytic()."""
    I, Z = render(width, height, T_cam, seed, frame_id, nan_fraction, hole, K)
    return sensor_from_analytic(I, Z, seed, frame_id, intensity_sigma, depth_noise, channels)


def raw_to_float(image, raw_z, depth_factor: float = DEPTH_FACTOR):
    """What frame ingest makes of a raw frame (benchmark_slam.cpp:60-80; the oracle's orc_ingest_* and the device's k_ingest do
    the same bit for bit, tests/test_gpu_parity.py::test_ingest_raw_frame_bit_exact): float32 grey 0..255 and float32 metres,
    (float) raw * (float)(1 / factor), with NaN where raw is 0.  Grey input only (BGR goes through the luma rule of ingest)."""
    assert image.ndim == 2
    scale = np.float32(1.0 / depth_factor)
    Z = raw_z.astype(np.float32) * scale
    Z[raw_z == 0] = np.nan
    return image.astype(np.float32), Z


def sensor_pair(width: int = 640, height: int = 480, xi_gt=XI_GT_PAIR, seed: int = SEED, frame_id: int = 0, **kw):
    """make_pair() in the sensor regime: ((grey_ref, raw_z_ref), (grey_cur, raw_z_cur), T_gt)"""
    T_gt = se3_exp(xi_gt)
    return (sensor_frame(width, height, None, seed, 2 * frame_id, **kw),
            sensor_frame(width, height, T_gt, seed, 2 * frame_id + 1, **kw), T_gt)


def make_pair(width: int = 640, height: int = 480, xi_gt=XI_GT_PAIR, seed: int = SEED, frame_id: int = 0):
    """Reference frame at the scene origin, current frame at exp(xi_gt).  The expected DenseTracker result
    (cur <- ref convention, dense_tracking.cpp:371) is T = exp(xi_gt)."""
    T_gt = se3_exp(xi_gt)
    ref = render(width, height, None, seed, 2 * frame_id)
    cur = render(width, height, T_gt, seed, 2 * frame_id + 1)
    return ref, cur, T_gt


def stream_poses(n_frames: int, xi_step=XI_STEP_STREAM):
    """Config 4: camera pose of frame t is exp(t * xi_step)."""
    return [se3_exp(np.asarray(xi_step) * t) for t in range(n_frames)]


def loop_closure_poses(n_candidates: int = 32, seed: int = SEED + 2, max_trans: float = 0.15, max_rot_deg: float = 5.0):
    """Config 5: candidate frames at hashed poses around the keyframe."""
    out = []
    for i in range(n_candidates):
        hs = [_hash_u32(np.array([i]), np.array([k]), seed)[0] / 4294967296.0 * 2 - 1 for k in range(8)]
        ups = np.array(hs[0:3])
        ups = ups / max(np.linalg.norm(ups), 1e-9) * max_trans * abs(hs[6])
        om = np.array(hs[3:6])
        om = om / max(np.linalg.norm(om), 1e-9) * np.deg2rad(max_rot_deg) * abs(hs[7])
        out.append(se3_exp(np.concatenate([ups, om])))
    return out


def loop_closure_scenario(width: int = 640, height: int = 480, n_candidates: int = 6, seed: int = SEED, decoys: bool = True,
                          pose_error: float = 0.1, sensor: bool = False):
    """Config 5 as a validator workload: one keyframe (id 100, pose = identity) and candidate keyframes at the hashed poses of
    loop_closure_poses().  Every entry: dict(id, frame=(I, Z), pose_true, pose) where `pose` is the pose the pose graph
    believes (the true one with `pose_error` of its twist removed, mirroring the 10 % perturbation of SURVEY.md config 5).
    With decoys, three candidates that validation must throw out are appended: a neighbour in id (odometry constraint),
    a frame of a different scene (seed + 77), and a frame without any depth (NaN result)."""
    def frame(T, sd, fid):  # sensor=True: the frame as the sensor delivers it, ingested to float planes (raw_to_float)
        return raw_to_float(*sensor_frame(width, height, T, sd, fid)) if sensor else render(width, height, T, sd, fid)

    key = dict(id=100, frame=frame(None, seed, 0), pose_true=np.eye(4), pose=np.eye(4))
    cands = []
    for i, T in enumerate(loop_closure_poses(n_candidates)):
        xi = se3_log(T)
        cands.append(dict(id=2 * i, frame=frame(T, seed, 1 + i), pose_true=T, pose=se3_exp(xi * (1.0 - pose_error))))
    if decoys:
        T = se3_exp(XI_STEP_STREAM)
        cands.append(dict(id=101, frame=frame(T, seed, 50), pose_true=T, pose=T))
        T = se3_exp(XI_GT_PAIR)
        cands.append(dict(id=60, frame=frame(T, seed + 77, 51), pose_true=T, pose=T))
        I, Z = frame(T, seed, 52)
        cands.append(dict(id=70, frame=(I, np.full_like(Z, np.nan)), pose_true=T, pose=T))
    return key, cands
