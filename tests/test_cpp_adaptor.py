"""The header-only C++ adaptor (include/dvo_amd/dense_tracking.hpp) mirrors dvo::DenseTracker / RgbdImagePyramid.
CPU: it compiles as plain C++11 against the C ABI (no Eigen / OpenCV in this image: the stand-in value types are used).
GPU: a caller written like dvo_ros' camera_dense_tracking.cpp gets the same pose as the Python binding."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "_build", "adaptor_example")


def _compile():
    from dvo_slam_amd import _build

    _build.build()
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    libdir = os.path.join(ROOT, "dvo_slam_amd")
    cmd = ["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "adaptor_example.cpp"), "-o", EXE, "-L" + libdir, "-ldvo_amd",
           "-Wl,-rpath," + libdir, "-Wl,--allow-shlib-undefined"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    return EXE


def test_adaptor_compiles_as_plain_cxx11():
    exe = _compile()
    assert os.path.exists(exe)
    # also with every warning a strict downstream build would enable, syntax only
    res = subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Wpedantic", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"),
                          os.path.join(ROOT, "examples", "adaptor_example.cpp")], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr


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
