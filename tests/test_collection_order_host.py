"""The driver runs `pytest tests -x -q -m gpu`: one failure ends the run, so WHAT RUNS FIRST decides what a bad day still proves.
tests/conftest.py orders the GPU files (pytest_collection_modifyitems): core parity of the hot path first, everything that starts processes
last, the bench subprocesses at the very end.  This test pins that order on the real collection (no GPU needed: --collect-only)."""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))


def _collected_files():
    r = subprocess.run([sys.executable, "-m", "pytest", "tests", "-m", "gpu", "--collect-only", "-q", "-p", "no:cacheprovider"], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    files = []
    for line in r.stdout.splitlines():
        if "::" in line:
            f = Path(line.split("::")[0]).stem
            if not files or files[-1] != f:
                files.append(f)
    return files


def test_gpu_files_run_core_parity_first_and_the_process_spawning_files_last():
    import conftest
    files = _collected_files()
    assert len(files) == len(set(files)), f"a file's tests are not contiguous in the run: {files}"
    listed = [f for f in files if f in conftest.GPU_FILE_ORDER]
    assert listed == [f for f in conftest.GPU_FILE_ORDER if f in files], files
    assert files[0] == "test_gpu_parity" and files[1] == "test_gpu_state", files[:3]
    assert files[-1] == "test_gpu_bench" and files[-4:] == ["test_gpu_partitioned", "test_gpu_dist_abi", "test_gpu_multi_device", "test_gpu_bench"], files[-5:]
    # every GPU test file on disk is either in the list or lands before the multi-process block
    on_disk = {p.stem for p in (ROOT / "tests").glob("test_gpu_*.py")}
    assert on_disk <= set(files), sorted(on_disk - set(files))
    multi = files.index("test_gpu_partitioned")
    for f in files:
        if f not in conftest.GPU_FILE_ORDER:
            assert files.index(f) < multi, f"{f} is not in GPU_FILE_ORDER and runs inside / after the multi-process block"
    assert set(conftest.GPU_FILE_ORDER) <= on_disk, sorted(set(conftest.GPU_FILE_ORDER) - on_disk)


def test_the_gpu_fixture_and_smoke_refuse_the_emulator_without_the_flag(tmp_path):
    """a library that exports `simt_counters` is the SIMT emulator of tests/simt: the `gpu` fixture's report and smoke() refuse it unless
    SIMT_ALLOW=1 (set only by the emulated children) — an environment variable alone cannot turn a GPU run green without a GPU"""
    import os
    built = ROOT / "tests" / "simt" / "_build" / "libsublinear_hip_simt.so"
    if not built.exists():
        r = subprocess.run([sys.executable, str(ROOT / "tests" / "simt" / "build.py")], capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0, r.stderr[-2000:]
    env = {k: v for k, v in os.environ.items() if k not in ("SIMT_ALLOW",)}
    env["SUBLINEAR_HIP_LIB"] = str(built)
    code = "import __graft_entry__ as g; g.smoke()"
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and "SIMT emulator" in r.stderr, r.stderr[-800:]
    code = "import sys; sys.path.insert(0, 'tests'); import conftest; print(conftest.loaded_library_report())"
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and "SIMT emulator" in r.stderr, r.stderr[-800:]
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600, env=dict(env, SIMT_ALLOW="1"))
    assert r.returncode == 0 and "[SIMT EMULATOR]" in r.stdout and "library:" in r.stdout, r.stdout + r.stderr[-800:]
    # the box's snapshot leaves the emulator build behind
    assert "tests/simt/_build/" in (ROOT / ".gpurunignore").read_text().split()


def test_gpu_suite_keeps_its_own_clock(tmp_path):
    """with the budget spent, waiting GPU tests are skipped with a reason instead of running into the driver's step limit"""
    import os
    t = tmp_path / "test_budget_probe.py"
    t.write_text("import pytest, time\npytestmark = pytest.mark.gpu\n"
                 "def test_first():\n    time.sleep(1.2)\n"
                 "def test_second():\n    raise AssertionError('must have been skipped')\n")
    (tmp_path / "conftest.py").write_text((ROOT / "tests" / "conftest.py").read_text().replace("ROOT = Path(__file__).resolve().parent.parent", f"ROOT = Path({str(ROOT)!r})"))
    env = dict(os.environ, SL_GPU_SUITE_BUDGET_S="1")
    r = subprocess.run([sys.executable, "-m", "pytest", str(t), "-m", "gpu", "-q", "-rs", "-p", "no:cacheprovider"], cwd=tmp_path, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "1 passed, 1 skipped" in r.stdout and "GPU suite budget" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
