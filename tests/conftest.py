import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def unpack_maze(bits, side):
    side = int(side)
    return np.unpackbits(bits)[: side * side].reshape(side, side).astype(np.uint8)


@pytest.fixture(scope="session")
def golden_episodes():
    return np.load(os.path.join(GOLDEN, "episodes.npz"))


@pytest.fixture(scope="session")
def golden_edges():
    return np.load(os.path.join(GOLDEN, "edges.npz"))


@pytest.fixture(scope="session")
def golden_astar():
    return np.load(os.path.join(GOLDEN, "astar.npz"))
