import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """Vectors produced by running the reference's NumPy oracle (tests/golden/make_golden.py)."""
    here = os.path.join(ROOT, "tests", "golden")
    data = np.load(os.path.join(here, "golden.npz"))
    with open(os.path.join(here, "golden.json")) as f:
        meta = json.load(f)
    return data, meta


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.lib()
    return o


@pytest.fixture(autouse=True)
def _device_status_is_clean(request):
    """After every GPU test: no kernel launched through the default context may have reported a failure through the device
    status word (a tripped loop bound of the sample-queue kernels: the reference asserts inside its kernel, src/render.cpp:68-73)."""
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    from ascendpathtracing_amd import render
    render.check_device_status()
