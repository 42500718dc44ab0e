import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle

    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def synth():
    from dvo_slam_amd import synth as s

    return s


@pytest.fixture(scope="session")
def small_pair(synth):
    """160x120 synthetic pair with ground truth (fast on the CPU)."""
    (Ir, Zr), (Ic, Zc), Tgt = synth.make_pair(160, 120, xi_gt=synth.XI_GT_PAIR * 0.5)
    return (Ir, Zr), (Ic, Zc), Tgt, synth.intrinsics_for(160, 120)
