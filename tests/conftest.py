import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


def _has_gpu() -> bool:
    try:
        import ctypes
        from sublinear_time_solver_amd import _lib
        n = ctypes.c_int(0)
        _lib.load().sl_device_count(ctypes.byref(n))
        return n.value > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    if not _has_gpu():
        pytest.fail("no HIP device or libsublinear_hip.so missing: GPU tests have no fallback")
    return True
