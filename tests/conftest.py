import importlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")         # bench.py's setting (one hardware queue per batch in flight and to spare; read when HIP initialises)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def maps():
    """The reference's map fixtures (tests/golden/maps.npz, packed by tests/golden/make_fixtures.py)."""
    z = np.load(os.path.join(GOLDEN, "maps.npz"))
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def maps_meta():
    return json.load(open(os.path.join(GOLDEN, "maps_meta.json")))


@pytest.fixture(scope="session")
def known():
    """Reference outputs recorded in SURVEY.md section 8c / Appendix A (tests/golden/known_answers.json)."""
    return json.load(open(os.path.join(GOLDEN, "known_answers.json")))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc


def tile2048(aisle1):
    """aisle1 tiled to 2048x2048 = np.tile(a,(4,2))[:2048,:2048]."""
    return np.ascontiguousarray(np.tile(aisle1, (4, 2))[:2048, :2048])


@pytest.fixture(scope="session")
def lsdmod():
    """The product package (directory name has a hyphen, so import it through importlib)."""
    return importlib.import_module("linesegmentdetector-slam_amd")
