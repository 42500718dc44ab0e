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


def _compile(name="adaptor_example"):
    from dvo_slam_amd import _build

    _build.build()
    exe = os.path.join(ROOT, "examples", "_build", name)
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    libdir = os.path.join(ROOT, "dvo_slam_amd")
    cmd = ["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
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


def test_constraints_adaptor_compiles_as_plain_cxx11():
    """include/dvo_amd/constraints.hpp: dvo_slam::constraints::ConstraintProposalValidator and friends."""
    assert os.path.exists(_compile("validator_example"))
    res = subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Wpedantic", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"),
                          os.path.join(ROOT, "examples", "validator_example.cpp")], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr


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
