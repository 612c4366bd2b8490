"""The kernels under AddressSanitizer (CPU suite; test infrastructure).  The SIMT emulator makes device memory host memory; built with
`-fsanitize=address` and exact-size "device" allocations (tests/simt/build.py, SIMT_SANITIZE=1), a kernel that reads or writes one element
past a device buffer — which an MI355X, whose allocations are padded to pages, would forgive silently and two rounds of blind edits could
have introduced — stops the run with the kernel's source line.  Runs smoke() and a selection of the `-m gpu` parity tests; the last test
shows the check has teeth (an off-by-one planted in the emulated copy of one kernel is reported)."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
SIMT = ROOT / "tests" / "simt"
T = "tests/test_gpu_"


def _build(extra_env=None):
    env = dict(os.environ, SIMT_SANITIZE="1", **(extra_env or {}))
    r = subprocess.run([sys.executable, str(SIMT / "build.py")], capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stderr[-4000:]
    return Path(r.stdout.strip().splitlines()[-1])


def _env(lib, **extra):
    rt = subprocess.run(["/opt/rocm/lib/llvm/bin/clang++", "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    if not Path(rt).exists():
        pytest.skip("no shared ASAN runtime beside the compiler")
    env = dict(os.environ, SUBLINEAR_HIP_LIB=str(lib), SIMT_ALLOW="1", SIMT_THREADS=str(max(1, min(4, os.cpu_count() or 1))), SIMT_FAKE_TORCH="2", LD_PRELOAD=rt,
               ASAN_OPTIONS="detect_leaks=0:verify_asan_link_order=0:detect_stack_use_after_return=0:halt_on_error=1:abort_on_error=0",
               SL_RCCL_LIB=str(lib.parent / "librccl.so.1"), SL_COMM_TIMEOUT_MS="300000")
    for k in ("SL_COMM_TRANSPORT", "SL_COMM_HALO", "SL_PUSH_SMALL", "SL_QUERY_WIDE", "SL_PW_INDEX_ONLY", "SL_CG_FUSED_DOT"):
        env.pop(k, None)
    env.update(extra)
    return env


@pytest.fixture(scope="module")
def asan_lib():
    return _build()


def test_smoke_is_clean_under_address_sanitizer(asan_lib):
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=ROOT, capture_output=True, text=True, timeout=1500, env=_env(asan_lib))
    assert r.returncode == 0 and "smoke ok" in r.stdout and "AddressSanitizer" not in r.stderr, r.stdout[-1000:] + r.stderr[-3000:]


# the default CPU suite runs the fast half (about a minute); SIMT_FULL=1 adds the rest; tools/simt_asan_full.sh runs every `-m gpu` file
FAST = [T + "matrix_mutate.py", T + "walk.py::test_serial_stream_pipeline_equals_the_one_lane_kernel", T + "walk.py::test_row_table_changes_no_bit",
        T + "band_geometry.py::test_band_kernel_every_slice_run_shape[default]", T + "band_geometry.py::test_band_kernel_every_slice_run_shape[SL_BAND_SPW=1]",
        T + "panels.py::test_paced_thin_panels_and_empty_super_panels", T + "longrows.py::test_hub_columns_and_batched_sparse_rounds_bitwise",
        T + "cg.py::test_cg_kat_from_reference_test"]
REST = [T + "matrix_trait.py", T + "walk.py::test_serial_stream_equals_the_executed_reference",
        T + "panels.py::test_paced_runs_longer_than_four_ending_a_panel_followed_by_the_same_row", T + "longrows.py::test_long_rows_spmv_neumann_both_orders",
        T + "fuzz.py::test_random_systems_bitwise[2049-300-17-False]", T + "order_any.py::test_order_any_on_uniform_columns[20000-8-5]",
        T + "band_geometry.py", T + "state.py", T + "walk.py::test_walk_values_bitwise_vs_oracle", T + "panels.py::test_paced_ragged_rows_hubs_duplicates",
        T + "panels.py::test_ragged_rows_hubs_duplicates_and_dense_push_rounds", T + "fuzz.py::test_paced_panels_random_structures_bitwise[4097-4097-cluster-12-True-2]",
        T + "mpass.py::test_row_slice_of_a_wide_band", T + "cg.py::test_cg_matches_oracle", T + "southwell.py",
        T + "session.py::test_session_survives_flooding_and_round_limit", T + "session.py::test_small_rounds_in_one_workgroup_give_the_same_answers"]
SELECTION = FAST + (REST if os.environ.get("SIMT_FULL") == "1" else [])


def test_parity_tests_are_clean_under_address_sanitizer(asan_lib):
    """a cross-section of the kernels: the round-6 mutators and serial walk, the matrix trait, the state object, paced panels with runs and
    empty super-panels, hub rows and hub columns through the sparse launch train, the order-free stream, a wide band's row slice, CG, Southwell"""
    have = subprocess.run([sys.executable, "-m", "pytest", "--collect-only", "-q", "-m", "gpu", "-p", "no:cacheprovider", *SELECTION], cwd=ROOT, capture_output=True, text=True, timeout=600)
    missing = [s for s in SELECTION if "::" in s and s.split("::")[1].split("[")[0] not in have.stdout]
    sel = [s for s in SELECTION if s not in missing]                       # (a renamed test must not turn this file red: it only narrows the selection)
    assert len(sel) >= len(SELECTION) - 3, missing
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-p", "no:cacheprovider", "--timeout", "900", *sel], cwd=ROOT, capture_output=True, text=True,
                       timeout=3000, env=_env(asan_lib))
    assert r.returncode == 0 and " passed" in r.stdout and " failed" not in r.stdout and "AddressSanitizer" not in r.stdout + r.stderr, r.stdout[-2500:] + r.stderr[-2500:]


def test_smoke_is_clean_under_undefined_behaviour_sanitizer():
    """SIMT_SANITIZE=2: + UndefinedBehaviorSanitizer (shift counts, signed overflow, float -> integer conversions out of range, static array
    bounds; not `alignment`).  SIMT_FULL=1 only (a second sanitized build: 50 s); the whole selection under it: profiles/r06_simt_ubsan.txt"""
    if os.environ.get("SIMT_FULL") != "1":
        pytest.skip("runs with SIMT_FULL=1 (profiles/r06_simt_ubsan.txt holds this round's sweep)")
    env0 = dict(os.environ, SIMT_SANITIZE="2")
    r = subprocess.run([sys.executable, str(SIMT / "build.py")], capture_output=True, text=True, timeout=1500, env=env0)
    assert r.returncode == 0, r.stderr[-4000:]
    lib = Path(r.stdout.strip().splitlines()[-1])
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=ROOT, capture_output=True, text=True, timeout=1500,
                       env=_env(lib, UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1"))
    assert r.returncode == 0 and "smoke ok" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stdout[-1000:] + r.stderr[-3000:]


def test_a_planted_off_by_one_is_reported(tmp_path):
    """teeth: sl_scale_array_kernel made to touch element n (one past the array) in a scratch build — ASAN names the kernel"""
    patch = tmp_path / "patch.py"
    patch.write_text("def patch(name, text):\n"
                     "    if name != 'sl_matrix.hip':\n        return text\n"
                     "    old = 'k < n; k += (uint64_t)gridDim.x * 256) p[k] = __dmul_rn(p[k], factor);'\n"
                     "    assert text.count(old) == 1\n"
                     "    return text.replace(old, 'k <= n; k += (uint64_t)gridDim.x * 256) p[k] = __dmul_rn(p[k], factor);')\n")
    lib = _build({"SIMT_PATCH": str(patch), "SIMT_BUILD": str(tmp_path / "build")})
    code = ("import numpy as np, sublinear_time_solver_amd as S\n"
            "from sublinear_time_solver_amd import generators as G\n"
            "rp, ci, va, b = G.sdd_rows(3000, 8, seed=1)\n"
            "m = S.SparseMatrix.from_csr(rp, ci, va, 3000, 3000, keep_csr=True)\n"
            "m.scale(2.0)\nprint('not reported')\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=900, env=_env(lib))
    assert r.returncode != 0 and "AddressSanitizer" in r.stderr and "heap-buffer-overflow" in r.stderr and "sl_scale_array_kernel" in r.stderr, r.stdout[-500:] + r.stderr[-3000:]
