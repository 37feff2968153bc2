import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True, scope="session")
def _session_kernel_variant():
    """DENSITY_TEST_VARIANT=<bits>: the whole session on another kernel family (density_hip_set_kernel_variant) — e.g. 2048, the other rotation
    encoder, for the test files that do not parametrise variants themselves.  (tests/test_gpu_chameleon.py sets its own per test.)"""
    v = os.environ.get("DENSITY_TEST_VARIANT")
    if v and _have_gpu():
        from density_amd import container
        container.set_kernel_variant(int(v))
    yield
