import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def transitions():
    import numpy as np
    d = np.load(os.path.join(GOLDEN, "reference_transitions.npz"))
    return {k: d[k] for k in d.files}


@pytest.fixture(scope="session")
def planner_vectors():
    import numpy as np
    d = np.load(os.path.join(GOLDEN, "planner_vectors.npz"))
    return {k: d[k] for k in d.files}


@pytest.fixture(scope="session")
def reacher_model():
    from trajopt_amd.model import get_model
    return get_model("reacher_7dof")


@pytest.fixture(scope="session")
def oracle_sim(reacher_model):
    import oracle
    oracle.build()
    from oracle.sim import OracleSim
    return OracleSim(reacher_model)
