import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def device():
    import mgr_amd  # noqa: F401
    from mgr_amd._capi import Device, device_count
    if device_count() < 1:
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box (there is no CPU fallback)")
    d = Device(0)
    yield d
    d.close()
