import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import mgr_amd  # noqa: E402,F401  (first: sizes the BLAS pools to the CPU quota before a test module imports numpy - _hostenv.py)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes minutes (CPU oracle over a whole decode set)")
    # The 2-rank data-parallel GPU tests (test_gpu_dataparallel.py) fork their rank processes from multiprocessing's fork
    # server.  It has to be started HERE, before anything in this process initialises the GPU: a process that has done so
    # must not exec another program, and a forked copy of it is no place to start a second GPU context either.
    if "not gpu" not in (config.getoption("-m", default="") or ""):
        from multiprocessing import forkserver
        forkserver.ensure_running()


@pytest.fixture(scope="session")
def device():
    import mgr_amd  # noqa: F401
    from mgr_amd._capi import Device, device_count
    if device_count() < 1:
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box (there is no CPU fallback)")
    d = Device(0)
    yield d
    d.close()
