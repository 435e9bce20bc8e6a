import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ml-qem_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def lima_backend():
    from blackwater.data.backends import StaticBackend

    return StaticBackend.from_json(os.path.join(GOLDEN, "fake_lima_backend_props.json"))


@pytest.fixture(scope="session")
def lima_props(lima_backend):
    from blackwater.data.utils import get_backend_properties_v1

    return get_backend_properties_v1(lima_backend)


@pytest.fixture(scope="session")
def g1():
    """The 300 graphs / labels / circuits behind goldens G1-G3."""
    import json

    import numpy as np

    z = dict(np.load(os.path.join(GOLDEN, "g1_dataset.npz")))
    with open(os.path.join(GOLDEN, "g1_circuits.json")) as fh:
        z["qasm"] = json.load(fh)
    return z
