"""The header-only C++ adaptor (include/dvo_amd/dense_tracking.hpp) mirrors dvo::DenseTracker / RgbdImagePyramid.
CPU: it compiles as plain C++11 against the C ABI (no Eigen / OpenCV in this image: the stand-in value types are used).
GPU: a caller written like dvo_ros' camera_dense_tracking.cpp gets the same pose as the Python binding."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "_build", "adaptor_example")


VEXE = os.path.join(ROOT, "examples", "_build", "validator_example")


MOCKS = os.path.join(ROOT, "tests", "mock_include")  # TEST-ONLY stand-ins for <Eigen/Geometry> and <opencv2/core/core.hpp>


def _compile(name="adaptor_example", mocks=False):
    """mocks=True: tests/mock_include in front of the include path -- the adaptor then takes its DVO_AMD_HAVE_EIGEN /
    DVO_AMD_HAVE_OPENCV branch (Eigen::Affine3d, cv::Mat signatures) instead of the plain-C++ stand-in types"""
    from dvo_slam_amd import _build

    _build.build()
    exe = os.path.join(ROOT, "examples", "_build", name + ("_mock" if mocks else ""))
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    libdir = os.path.join(ROOT, "dvo_slam_amd")
    # include/dvo_amd_compat in front: <dvo/dense_tracking.h>, <dvo/core/rgbd_image.h>, <dvo/core/point_selection.h> resolve
    # to the forwarding headers, as they would in a dvo_slam build with the include path switched
    cmd = ["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-pthread"] + (["-I" + MOCKS] if mocks else []) + [
           "-I" + os.path.join(ROOT, "include", "dvo_amd_compat"),
           "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", name + ".cpp"), "-o", exe, "-L" + libdir, "-ldvo_amd",
           "-Wl,-rpath," + libdir, "-Wl,--allow-shlib-undefined"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    return exe


def test_adaptor_compiles_as_plain_cxx11():
    exe = _compile()
    assert os.path.exists(exe)
    # also with every warning a strict downstream build would enable, syntax only
    res = subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Wpedantic", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"),
                          os.path.join(ROOT, "examples", "adaptor_example.cpp")], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr


@pytest.mark.parametrize("name", ["adaptor_example", "boundary_callsites", "local_tracker_example", "validator_example",
                                  "reference_signatures"])
def test_production_branch_of_the_adaptor_compiles_against_eigen_and_opencv_mocks(name):
    """the branch of the adaptor that carries the reference's REAL signatures --
    bool match(..., Eigen::Affine3d&), create(const cv::Mat&, const cv::Mat&), cv::Mat computeIntensityErrorImage
    (dvo_core/include/dvo/dense_tracking.h:156-162, core/rgbd_image.h:135) -- had never been through a compiler: neither library
    exists in this image.  tests/mock_include holds test-only minimal <Eigen/Geometry> and <opencv2/core/core.hpp> exposing
    exactly the members the adaptor uses; every example, and examples/reference_signatures.cpp which calls those signatures and
    static_asserts their types, compiles against them with -std=c++11 -Wall -Wextra -Werror.  A compile proof against mocks,
    not against the real libraries (INTEGRATION.md says so)."""
    exe = _compile(name, mocks=True)
    assert os.path.exists(exe)
    # the branch was really taken: the translation unit sees Eigen's type behind dvo::core::AffineTransformd
    probe = ('#include "dvo_amd/dense_tracking.hpp"\n#include <type_traits>\n'
             'static_assert(std::is_same<dvo::core::AffineTransformd, Eigen::Affine3d>::value, "Eigen branch");\n'
             '#if !defined(DVO_AMD_HAVE_OPENCV)\n#error no OpenCV branch\n#endif\nint main() { return 0; }\n')
    res = subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-I" + MOCKS, "-I" + os.path.join(ROOT, "include"), "-x", "c++", "-"],
                         input=probe, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    if name == "boundary_callsites":  # needs no GPU: the Eigen-typed statistics dump runs
        res = subprocess.run([exe], capture_output=True, text=True)
        assert res.returncode == 0 and "boundary call sites ok" in res.stdout, res.stderr


def test_constraints_adaptor_compiles_as_plain_cxx11():
    """include/dvo_amd/constraints.hpp: dvo_slam::constraints::ConstraintProposalValidator and friends."""
    assert os.path.exists(_compile("validator_example"))
    res = subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Wpedantic", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"),
                          os.path.join(ROOT, "examples", "validator_example.cpp")], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr


def test_local_tracker_written_like_the_reference_compiles_against_the_forwarding_headers():
    """examples/local_tracker_example.cpp is the body of dvo_slam/src/local_tracker.cpp:40-74,127-213 with its original
    includes (<dvo/dense_tracking.h>, <dvo/core/point_selection.h>, <dvo/core/rgbd_image.h>): PointSelection, the
    match(PointSelection&, ...) overloads, level(i).buildPointCloud() and computeIntensityErrorImage all resolve."""
    assert os.path.exists(_compile("local_tracker_example"))
    text = open(os.path.join(ROOT, "examples", "local_tracker_example.cpp")).read()
    for needle in ("#include <dvo/dense_tracking.h>", "#include <dvo/core/point_selection.h>", "match(*keyframe_points_, *frame, r_odometry)",
                   "tracker->match(*ref, *cur, *r)", "keyframe_points_.swap(active_frame_points_)", "buildAccelerationStructure()"):
        assert needle in text, needle
    for fwd in ("dvo/dense_tracking.h", "dvo/core/rgbd_image.h", "dvo/core/point_selection.h", "dvo/core/intrinsic_matrix.h"):
        assert os.path.exists(os.path.join(ROOT, "include", "dvo_amd_compat", fwd)), fwd


def test_reference_call_sites_that_round_2_broke_compile_and_run():
    """examples/boundary_callsites.cpp holds the bodies of dvo_ros' updateConfigFromDynamicReconfigure (configtools.h:32-82: every
    Config field, the five dead ones and the two weight_calculation.h enums included) and of the statistics dump in
    keyframe_graph.cpp:364-371 (stream operators of Stats / LevelStats / IterationStats, non-const LastIterationWithIncrement(),
    InformationConditionNumber()) as the reference writes them, plus the includes <dvo/core/surface_pyramid.h> and
    <dvo/core/point_selection_predicates.h>.  It needs no GPU: compiled with -Werror and RUN here."""
    exe = _compile("boundary_callsites")
    res = subprocess.run([exe], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    assert "boundary call sites ok" in res.stdout
    # the statistics went through the reference's stream format (dense_tracking.h:239-291)
    assert "2 levels" in res.stderr and "Termination: LogLikelihoodDecreased" in res.stderr and "condition number, finest level: " in res.stderr
    assert "Iteration: 2 ValidConstraints: 3898 DataLogLikelihood: -10002 PriorLogLikelihood: 0" in res.stderr
    for fwd in ("dvo/core/surface_pyramid.h", "dvo/core/point_selection_predicates.h", "dvo/core/weight_calculation.h"):
        assert os.path.exists(os.path.join(ROOT, "include", "dvo_amd_compat", fwd)), fwd
    text = open(os.path.join(ROOT, "examples", "local_tracker_example.cpp")).read()
    assert "#include <dvo/core/point_selection_predicates.h>" in text  # local_tracker.cpp:24, dropped in round 2


CEXE = os.path.join(ROOT, "examples", "_build", "c_abi_example")


def _compile_c():
    from dvo_slam_amd import _build

    _build.build()
    os.makedirs(os.path.dirname(CEXE), exist_ok=True)
    libdir = os.path.join(ROOT, "dvo_slam_amd")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Wpedantic", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "c_abi_example.c"), "-o", CEXE, "-L" + libdir, "-ldvo_amd",
           "-Wl,-rpath," + libdir, "-Wl,--allow-shlib-undefined"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    return CEXE


def test_c_abi_header_is_plain_c99():
    """include/dvo_amd.h from a C translation unit (no C++ anywhere): the binding a C host would write."""
    assert os.path.exists(_compile_c())


@pytest.mark.gpu
def test_c_caller_with_raw_frames_matches_python_binding(tmp_path, synth):
    from dvo_slam_amd import capi

    exe = CEXE if os.path.exists(CEXE) else _compile_c()
    w, h = 640, 480
    (Ir, Zr), (Ic, Zc), Tgt = synth.make_pair(w, h)
    K = synth.intrinsics_for(w, h)
    raw_r, raw_c = synth.to_raw(Ir, Zr), synth.to_raw(Ic, Zc)
    paths = []
    for name, arr in (("rb", raw_r[0]), ("rz", raw_r[1]), ("cb", raw_c[0]), ("cz", raw_c[1])):
        p = tmp_path / (name + ".raw")
        np.ascontiguousarray(arr).tofile(p)
        paths.append(str(p))
    res = subprocess.run([exe, str(w), str(h)] + [repr(float(k)) for k in K] + paths, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    lines = res.stdout.strip().splitlines()
    assert lines[0].startswith("isnan 0")
    T = np.array([[float(v) for v in ln.split()] for ln in lines[1:5]])
    trk = capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=0))
    ref = trk.match(capi.RgbdImagePyramid.from_raw(*raw_r, K, 4), capi.RgbdImagePyramid.from_raw(*raw_c, K, 4))
    assert synth.pose_error(ref.Transformation, T) <= 1e-9  # same library, same inputs
    assert synth.pose_error(Tgt, T) < 2e-3  # 8-bit / 0.2 mm quantisation of the raw frames
    assert lines[5].startswith("1305031102.175303936 ") and len(lines[5].split()) == 8


@pytest.mark.gpu
def test_adaptor_matches_python_binding(tmp_path, synth):
    from dvo_slam_amd import capi

    exe = EXE if os.path.exists(EXE) else _compile()
    w, h = 640, 480
    (Ir, Zr), (Ic, Zc), Tgt = synth.make_pair(w, h)
    K = synth.intrinsics_for(w, h)
    paths = []
    for name, arr in (("ri", Ir), ("rz", Zr), ("ci", Ic), ("cz", Zc)):
        p = tmp_path / (name + ".f32")
        np.ascontiguousarray(arr, dtype=np.float32).tofile(p)
        paths.append(str(p))
    env = dict(os.environ)
    # the executable must resolve the same HIP runtime the Python process would (torch bundles one): use the system one
    res = subprocess.run([exe, str(w), str(h)] + [repr(float(k)) for k in K] + paths, capture_output=True, text=True, env=env)
    assert res.returncode == 0, res.stderr
    lines = res.stdout.strip().splitlines()
    assert lines[0].startswith("isnan 0")
    T = np.array([[float(v) for v in ln.split()] for ln in lines[1:5]])
    trk = capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=0))
    ref = trk.match(capi.RgbdImagePyramid(Ir, Zr, K, 4), capi.RgbdImagePyramid(Ic, Zc, K, 4))
    assert synth.pose_error(ref.Transformation, T) <= 1e-9  # same library, same inputs
    assert synth.pose_error(Tgt, T) < 2e-5
    assert sum(1 for ln in lines if ln.startswith("level ")) == 4


@pytest.mark.gpu
def test_constraints_adaptor_matches_python_binding(tmp_path, synth):
    """A caller written like keyframe_graph.cpp (C++ adaptor) keeps the same constraints as the Python mirror."""
    from dvo_slam_amd import capi, constraints as Cn

    exe = VEXE if os.path.exists(VEXE) else _compile("validator_example")
    w, h = 640, 480
    K = synth.intrinsics_for(w, h)
    key, cands = synth.loop_closure_scenario(w, h, 4)
    entries = [key] + cands
    with open(tmp_path / "frames.txt", "w") as fh:
        for e in entries:
            fh.write(str(e["id"]) + " " + " ".join(repr(float(v)) for v in np.asarray(e["pose"]).reshape(-1)) + "\n")
            np.ascontiguousarray(e["frame"][0], dtype=np.float32).tofile(tmp_path / f"{e['id']}_i.f32")
            np.ascontiguousarray(e["frame"][1], dtype=np.float32).tofile(tmp_path / f"{e['id']}_z.f32")
    thresholds = dict(min_constraint_ratio=0.2, ratio_coarse=-1e300, ratio_fine=-1e300)
    res = subprocess.run([exe, str(tmp_path), str(w), str(h)] + [repr(float(k)) for k in K] +
                         [repr(thresholds["min_constraint_ratio"]), "-1e300", "-1e300"], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    lines = res.stdout.strip().splitlines()
    got = []
    for ln in lines[1:]:
        t = ln.split()
        got.append((int(t[1]), int(t[2]), float(t[4]), int(t[6]), int(t[8]), np.array([float(v) for v in t[9:25]]).reshape(4, 4)))
    # the same scenario through the Python mirror (evaluation seeded the same way: every frame against itself)
    trk = capi.DenseTracker(capi.Config())

    def mk(e):
        p = capi.RgbdImagePyramid(e["frame"][0], e["frame"][1], K, 4)
        return Cn.Keyframe(e["id"], p, e["pose"], Cn.LogLikelihoodTrackingResultEvaluation(trk.match(p, p)))

    kfs = [mk(e) for e in entries]
    want = Cn.createConstraintProposalValidator(**thresholds).validate(Cn.proposalsForCandidates(kfs[0], kfs[1:]))
    assert int(lines[0].split()[1]) == len(want) == len(got) > 0
    for g, p in zip(got, want):
        assert (g[0], g[1]) == (p.Reference.id, p.Current.id)
        assert g[3] == len(p.Votes) and g[4] == int(p.Accept())
        assert abs(g[2] - p.TotalScore()) <= 1e-9 * max(1.0, abs(p.TotalScore()))
        assert synth.pose_error(g[5], p.TrackingResult.Transformation) <= 1e-9  # same library, same inputs


@pytest.mark.gpu
def test_local_tracker_example_equals_the_python_binding(tmp_path, synth):
    """The LocalTracker written like the reference (two trackers, two PointSelections sharing one predicate with non-default
    gradient thresholds, two threads) against the same sequence of matches issued through the Python binding."""
    from dvo_slam_amd import capi

    exe = _compile("local_tracker_example")
    w, h, n = 640, 480, 7
    K = synth.intrinsics_for(w, h)
    poses = synth.stream_poses(n)
    frames = [synth.render(w, h, poses[t], frame_id=t) for t in range(n)]
    for i, (I, Z) in enumerate(frames):
        np.ascontiguousarray(I, dtype=np.float32).tofile(tmp_path / f"{i}_i.f32")
        np.ascontiguousarray(Z, dtype=np.float32).tofile(tmp_path / f"{i}_z.f32")
    ti, td = 2.5, 0.01
    res = subprocess.run([exe, str(tmp_path), str(n), str(w), str(h)] + [repr(float(k)) for k in K] + [repr(ti), repr(td)],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    got = {}
    newmaps = []
    for ln in res.stdout.strip().splitlines():
        t = ln.split()
        if t[0] in ("odometry", "keyframe"):
            assert t[3] == "0"
            got[(t[0], int(t[1]))] = np.array([float(v) for v in t[4:20]]).reshape(4, 4).T
        elif t[0] == "newmap":
            newmaps.append(int(t[2]))
    # the same front end through the Python binding
    cfg = capi.Config(UseInitialEstimate=True, IntensityDerivativeThreshold=ti, DepthDerivativeThreshold=td)
    trk = capi.DenseTracker(cfg)
    pyr = [capi.RgbdImagePyramid(I, Z, K, 4) for I, Z in frames]
    key, last = 0, 1
    last_keyframe_pose = trk.match(pyr[0], pyr[1], np.eye(4)).Transformation
    want_newmaps = []
    for i in range(2, n):
        r_key = trk.match(pyr[key], pyr[i], np.linalg.inv(last_keyframe_pose))
        r_odo = trk.match(pyr[last], pyr[i], np.eye(4))
        assert synth.pose_error(r_key.Transformation, got[("keyframe", i)]) <= 1e-9
        assert synth.pose_error(r_odo.Transformation, got[("odometry", i)]) <= 1e-9
        L = r_key.Levels[-1]
        ok = np.linalg.norm(r_key.Transformation[:3, 3]) <= 0.02 and L["Iterations"][-1]["ValidConstraints"] / L["ValidPixels"] >= 0.3
        want_newmaps.append(0 if ok else 1)
        if ok:
            last_keyframe_pose, last = r_key.Transformation, i
        else:
            key, last, last_keyframe_pose = last, i, r_odo.Transformation
    assert newmaps == want_newmaps and sum(newmaps) >= 1  # the keyframe was swapped at least once (selection swap exercised)
    sel = [ln.split() for ln in res.stdout.splitlines() if ln.startswith("selection")][0]
    count, mask = pyr[0].select(1, ti, td)
    assert int(sel[4]) == int(sel[8]) == count and int(sel[6]) == (w * h) // 4
    z = pyr[0].plane(1, 1)
    assert abs(float(sel[10]) - float(z.ravel()[mask.ravel().astype(bool)].astype(np.float64).sum())) <= 1e-3 * count
    err = [ln.split() for ln in res.stdout.splitlines() if ln.startswith("errorimage")][0]
    img = trk.computeIntensityErrorImage(pyr[0], pyr[1], np.eye(4), level=1)
    assert (int(err[1]), int(err[2])) == img.shape and abs(float(err[3]) - float(img.astype(np.float64).sum())) <= 1e-6 * img.size


@pytest.mark.gpu
def test_reference_signatures_run_on_the_gpu(tmp_path, synth):
    """examples/reference_signatures.cpp (Eigen::Affine3d / cv::Mat signatures, compiled against the functional test mocks) fed
    the sensor-regime frames as raw 8-bit grey + uint16 depth, the way benchmark_slam.cpp:56-80 feeds the reference: the pose it
    prints is the Python binding's for the same frames."""
    from dvo_slam_amd import capi

    exe = _compile("reference_signatures", mocks=True)
    w, h = 640, 480
    ref, cur, Tgt = synth.sensor_pair(w, h)
    K = synth.intrinsics_for(w, h)
    paths = []
    for name, arr in (("rg", ref[0]), ("rz", ref[1]), ("cg", cur[0]), ("cz", cur[1])):
        p = tmp_path / (name + ".raw")
        np.ascontiguousarray(arr).tofile(p)
        paths.append(str(p))
    res = subprocess.run([exe, str(w), str(h)] + [repr(float(k)) for k in K] + paths, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    lines = res.stdout.strip().splitlines()
    assert lines[0] == "success 1" and lines[5] == "selection success 1 same 1"
    T = np.array([[float(v) for v in ln.split()] for ln in lines[1:5]])
    trk = capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=0))
    want = trk.match(capi.RgbdImagePyramid(*synth.raw_to_float(*ref), K, 4), capi.RgbdImagePyramid(*synth.raw_to_float(*cur), K, 4))
    assert np.array_equal(want.Transformation, T)  # same library, same planes: the same bits (printed with 17 digits)
    assert synth.pose_error(Tgt, T) < 1e-3
    assert abs(float(lines[6].split()[1]) - np.linalg.inv(T)[0, 3]) < 1e-12
    assert lines[7].startswith("error_image 320 x 240 type_is_32f 1 sum ") and float(lines[7].split()[-1]) > 0
