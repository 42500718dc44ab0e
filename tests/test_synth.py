"""Host logic: the synthetic scene generator is deterministic and geometrically consistent."""
import numpy as np


def test_render_is_deterministic_and_in_contract(synth):
    a = synth.render(160, 120)
    b = synth.render(160, 120)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1], equal_nan=True)
    I, Z = a
    assert I.dtype == np.float32 and Z.dtype == np.float32 and I.shape == (120, 160)
    assert I.min() >= 0 and I.max() <= 255 and I.std() > 5
    assert 0.005 < np.isnan(Z).mean() < 0.1
    assert np.nanmin(Z) > 0.3 and np.nanmax(Z) <= 3.0 + 1e-5  # nothing beyond the far wall


def test_depth_is_geometrically_consistent_between_views(synth):
    """back-project a reference pixel, move it into the current camera, compare with the current depth map"""
    w, h = 160, 120
    (Ir, Zr), (Ic, Zc), T = synth.make_pair(w, h, xi_gt=synth.XI_GT_PAIR)
    fx, fy, ox, oy = [float(k) for k in synth.intrinsics_for(w, h)]
    Ti = np.linalg.inv(T)
    err = []
    for (v, u) in [(30, 40), (60, 80), (90, 120), (100, 20)]:
        z = Zr[v, u]
        if np.isnan(z):
            continue
        p = np.array([(u - ox) / fx * z, (v - oy) / fy * z, z, 1.0])
        q = Ti @ p
        uc, vc = q[0] / q[2] * fx + ox, q[1] / q[2] * fy + oy
        zc = Zc[int(round(vc)), int(round(uc))]
        if not np.isnan(zc):
            err.append(abs(zc - q[2]))
    assert err and max(err) < 0.02


def test_pose_error_metric(synth):
    T = synth.se3_exp([0.01, 0, 0, 0, 0.02, 0])
    assert synth.pose_error(T, T) < 1e-15
    assert abs(synth.pose_error(np.eye(4), T) - np.linalg.norm([0.01, 0.02])) < 1e-12
    assert np.allclose(synth.se3_log(synth.se3_exp(synth.XI_GT_PAIR)), synth.XI_GT_PAIR, atol=1e-14)


def test_workload_generators(synth):
    poses = synth.stream_poses(5)
    assert len(poses) == 5 and np.allclose(poses[0], np.eye(4))
    assert np.allclose(synth.se3_log(poses[3]), 3 * synth.XI_STEP_STREAM, atol=1e-12)
    lc = synth.loop_closure_poses(32)
    assert len(lc) == 32
    for T in lc:
        xi = synth.se3_log(T)
        assert np.linalg.norm(xi[3:]) <= np.deg2rad(5.0) + 1e-9


def test_the_sensor_model_is_what_it_says(synth):
    """8-bit grey, uint16 depth on the 0.2 mm grid with 0 = no measurement, depth noise of the reference's own sigma"""
    (gray, raw_z), _, _ = synth.sensor_pair(640, 480)
    I, Z = synth.render(640, 480)
    assert gray.dtype == np.uint8 and raw_z.dtype == np.uint16 and gray.shape == raw_z.shape == (480, 640)
    assert np.array_equal(raw_z == 0, np.isnan(Z))
    If, Zf = synth.raw_to_float(gray, raw_z)
    m = ~np.isnan(Z)
    ratio = (Zf[m].astype(np.float64) - Z[m]) / synth.depth_std_dev(Z[m])
    assert abs(ratio.std() - 1.0) < 0.02 and abs(ratio.mean()) < 0.01          # N(0, sigma_z(z)), on top of +-0.1 mm rounding
    assert abs(synth.depth_std_dev(0.4) - 0.0012) < 1e-15 and abs(synth.depth_std_dev(2.4) - 0.0088) < 1e-15
    assert 1.3 < (If - I).std() < 1.7
    again = synth.sensor_frame(640, 480)  # deterministic
    assert np.array_equal(again[0], gray) and np.array_equal(again[1], raw_z)
    other = synth.sensor_frame(640, 480, frame_id=1)
    assert not np.array_equal(other[1], raw_z)
