import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


# Under the SIMT emulator (tests/test_simt_emulated.py sets SUBLINEAR_HIP_LIB to it) the tests that hand torch.cuda tensors to the ABI get
# tests/simt/fake_torch.py instead: host arrays whose address is a valid "device" address there.  Never in effect on a GPU box.
if "simt" in Path(os.environ.get("SUBLINEAR_HIP_LIB", "")).name and os.environ.get("SIMT_FAKE_TORCH") in ("1", "2"):
    import importlib.util
    if os.environ["SIMT_FAKE_TORCH"] == "2":      # the real torch, host tensors where a test asks for device ones (tests/simt/torch_on_host.py)
        _spec = importlib.util.spec_from_file_location("torch_on_host", str(ROOT / "tests" / "simt" / "torch_on_host.py"))
        _mod = importlib.util.module_from_spec(_spec)
        _spec.loader.exec_module(_mod)
        _mod.install()
    else:
        _spec = importlib.util.spec_from_file_location("torch", str(ROOT / "tests" / "simt" / "fake_torch.py"))
        sys.modules["torch"] = importlib.util.module_from_spec(_spec)
        _spec.loader.exec_module(sys.modules["torch"])


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (they are git-ignored): build the library and the oracle checker once."""
    import subprocess
    missing = [p for p in (ROOT / "sublinear_time_solver_amd" / "libsublinear_hip.so", ROOT / "oracle" / "liboracle.so",
                           ROOT / "oracle" / "liboracle_fast.so") if not p.exists()]
    if missing:
        jobs = str(max(1, min(8, os.cpu_count() or 1)))
        subprocess.run(["make", "-C", str(ROOT / "sublinear_time_solver_amd" / "csrc"), "-j", jobs], check=True, capture_output=True)
        subprocess.run(["make", "-C", str(ROOT / "oracle"), "liboracle.so", "liboracle_fast.so"], check=True, capture_output=True)


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


@pytest.fixture
def c_smoke_exe(tmp_path):
    """tests/c/abi_smoke.c built as strict C99 against include/sublinear_hip.h and linked to the library"""
    import subprocess
    exe = tmp_path / "abi_smoke"
    pkg = ROOT / "sublinear_time_solver_amd"
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", f"-I{ROOT / 'include'}", str(ROOT / "tests" / "c" / "abi_smoke.c"),
                        "-o", str(exe), f"-L{pkg}", "-lsublinear_hip", "-lm", f"-Wl,-rpath,{pkg}"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


@pytest.fixture
def node_surface_script():
    """bindings/node built (N-API addon, gcc); returns the path of the JavaScript surface test, or skips without node"""
    import shutil
    import subprocess
    if not shutil.which("node") or not Path("/usr/include/node/node_api.h").exists():
        pytest.skip("node / node_api.h not available")
    r = subprocess.run(["make", "-C", str(ROOT / "bindings" / "node")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return ROOT / "tests" / "js" / "surface_test.js"
