import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _npz(name):
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden_kat():
    return _npz("kat.npz")


@pytest.fixture(scope="session")
def golden_bvh():
    return _npz("bvh.npz")


@pytest.fixture(scope="session")
def golden_rays():
    return _npz("rays.npz")


@pytest.fixture(scope="session")
def golden_images():
    return _npz("images.npz")


@pytest.fixture(scope="session")
def golden_shade():
    return _npz("shade.npz")


@pytest.fixture(scope="session")
def gpu_device():
    """Fails (never skips) when the HIP library or the GPU is missing: the GPU tests
    must not pass on a machine where the native path did not run."""
    from raytracer_amd import engine
    lib = engine.host_lib()
    rc = lib.rtxh_select_device(0)
    assert rc == 0, "no usable GPU: " + lib.rtxh_last_error().decode()
    return 0
