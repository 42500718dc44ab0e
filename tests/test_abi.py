"""CPU tests of the C-ABI library: it loads, exports every symbol include/dvo_amd.h and include/dvo_amd_debug.h declare, its host-side helpers are
correct, and without a GPU every compute entry point fails loudly (there is no CPU fallback in the product path)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    from dvo_slam_amd import capi as c

    c.lib()
    return c


def _declared_functions(header="dvo_amd.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dvo_amd_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(capi):
    boundary, scaffolding = _declared_functions(), _declared_functions("dvo_amd_debug.h")
    # the boundary header holds no test probe, diagnostic or micro-benchmark; the debug header holds nothing else
    assert not [n for n in boundary if "_debug_" in n or "_bench_" in n or n == "dvo_amd_kernel_timing"]
    assert all("_debug_" in n or "_bench_" in n or n == "dvo_amd_kernel_timing" for n in scaffolding) and len(scaffolding) >= 10
    declared = sorted(set(boundary + scaffolding))
    assert len(boundary) >= 20
    L = capi.lib()
    missing = [name for name in declared if not hasattr(L, name)]
    assert not missing, missing
    assert sorted(capi.EXPORTS) == declared  # the Python binding covers the whole header
    assert L.dvo_amd_abi_version() == 3


def test_default_config_matches_reference_defaults(capi):
    c = capi.Config()  # dense_tracking_config.cpp:27-41
    assert (c.FirstLevel, c.LastLevel, c.MaxIterationsPerLevel) == (3, 1, 100)
    assert c.Precision == 5e-7 and c.Mu == 0.0 and c.UseInitialEstimate is False
    assert c.IntensityDerivativeThreshold == 0.0 and c.DepthDerivativeThreshold == 0.0
    assert c.getNumLevels() == 4 and c.IsSane()
    assert not capi.Config(FirstLevel=0, LastLevel=1).IsSane()


def test_status_strings(capi):
    L = capi.lib()
    assert L.dvo_amd_status_string(0) == b"ok"
    for s in range(1, 11):
        assert len(L.dvo_amd_status_string(s)) > 3
    assert b"no CPU fallback" in L.dvo_amd_status_string(2)


def test_host_se3_helpers_agree_with_oracle(capi, orc, synth):
    rng = np.random.default_rng(5)
    for scale in (1e-13, 1e-5, 0.05, 0.8):
        xi = rng.normal(size=6) * scale
        T = capi.se3_exp(xi)
        assert np.allclose(T, orc.se3_exp(xi), atol=1e-15)
        assert np.allclose(T, synth.se3_exp(xi), atol=1e-12)
        assert np.allclose(capi.se3_log(T), xi, rtol=1e-9, atol=1e-15)


def test_host_solve6(capi):
    rng = np.random.default_rng(6)
    for _ in range(5):
        M = rng.normal(size=(6, 6))
        A = M @ M.T + np.diag(rng.uniform(0, 1e3, 6))
        b = rng.normal(size=6)
        assert np.allclose(capi.solve6(A, b), np.linalg.solve(A, b), rtol=1e-10, atol=1e-12)


def test_compute_entry_points_fail_loudly_without_a_gpu(capi):
    if capi.lib().dvo_amd_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(capi.DvoAmdError) as e:
        capi.DenseTracker(capi.Config())
    assert e.value.status == 2  # DVO_AMD_ERR_NO_DEVICE
    I = np.zeros((16, 16), np.float32)
    with pytest.raises(capi.DvoAmdError) as e:
        capi.RgbdImagePyramid(I, I, (10, 10, 8, 8), 1)
    assert e.value.status == 2


def test_product_package_does_not_touch_the_oracle():
    """the product path may not import, link or call anything under oracle/"""
    pkg = os.path.join(ROOT, "dvo_slam_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "dvo_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f
    for f in os.listdir(os.path.join(ROOT, "include")):
        p = os.path.join(ROOT, "include", f)
        if os.path.isfile(p):
            assert "oracle" not in open(p).read().lower().replace("no oracle", "")


def test_build_id_ties_the_binary_to_the_sources(tmp_path):
    """dvo_amd_build_id() is the hash of csrc/*, include/dvo_amd.h and the compiler flags the library was built from; a library
    whose id is not the hash of the sources next to it is rebuilt (where hipcc exists) or refused -- never silently used
    (staleness used to be mtime-only)."""
    import subprocess
    import sys

    from dvo_slam_amd import _build, capi

    assert capi.build_id() == _build.source_id() == _build.library_id()
    assert len(capi.build_id()) == 16 and int(capi.build_id(), 16) >= 0
    # a copy whose embedded id was tampered with reads as stale ...
    blob = open(_build.LIB_PATH, "rb").read()
    at = blob.find(b"DVO_AMD_BUILD_ID=") + len(b"DVO_AMD_BUILD_ID=")
    stale = tmp_path / "libdvo_amd.so"
    stale.write_bytes(blob[:at] + b"0" * 16 + blob[at + 16:])
    assert _build.library_id(str(stale)) == "0" * 16 != _build.source_id()
    # ... and where it cannot be rebuilt the binding refuses it instead of running it
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from dvo_slam_amd import _build, capi\n"
            "_build.LIB_PATH = %r\n"
            "def no_compiler(*a, **k): raise RuntimeError('no hipcc on this box')\n"
            "_build.build = no_compiler\n"
            "try:\n    capi.lib()\nexcept RuntimeError as e:\n    print('REFUSED', e)\n" % (ROOT, str(stale)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "REFUSED" in out.stdout and "stale" in out.stdout, out.stdout + out.stderr
