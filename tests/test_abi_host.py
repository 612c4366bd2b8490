"""CPU-side checks: the C-ABI library loads and exports every symbol include/sublinear_hip.h declares,
no FMA contraction in the parity kernels, generators are deterministic, host logic validates input.
No compute calls (there is no GPU in the build container)."""
import ctypes
import re
import subprocess
from pathlib import Path

import numpy as np
import pytest

from sublinear_time_solver_amd import _lib as L
from sublinear_time_solver_amd import generators as G

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "sublinear_hip.h"


def declared_symbols():
    text = HEADER.read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sl_[a-z0-9_]+)\s*\(", text)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = L.load()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in sublinear_hip.h but not exported"
        assert n in L.SIGNATURES, f"{n} has no ctypes signature"
    assert lib.sl_abi_version() == 5
    assert lib.sl_status_string(3) == b"ConvergenceFailure"


def test_no_torch_or_cxx_types_in_the_abi():
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)     # declarations only
    assert "torch" not in text and "std::" not in text and "at::" not in text
    assert 'extern "C"' in text


# (the no-FMA-contraction check of the parity kernels lives in tests/test_isa_host.py, with the register / LDS / scratch pins)


def test_sdd_generator_properties():
    n, k = 4096, 16
    for w in (0, 64):
        rp, ci, va, b = G.sdd_rows(n, k, seed=1, half_bandwidth=w)
        assert rp[-1] == n * k and ci.size == n * k
        C = ci.reshape(n, k).astype(np.int64)
        V = va.reshape(n, k)
        assert (np.diff(C, axis=1) > 0).all(), "columns ascending and distinct"
        rows = np.arange(n)[:, None]
        isdiag = C == rows
        assert (isdiag.sum(axis=1) == 1).all()
        d = V[isdiag]
        off = np.abs(np.where(isdiag, 0.0, V)).sum(axis=1)
        assert (off <= d / 2 + 1e-12).all(), "strictly row dominant with margin 2"
        if w:
            assert (np.abs(C - rows) <= w).all()
        assert not np.allclose(V[1, 1], V[1, 2])
    # any row range reproduces the same rows
    a = G.sdd_rows(n, k, 3, 0, 100, 200)
    full = G.sdd_rows(n, k, 3, 0)
    assert (a[1] == full[1][100 * k:200 * k]).all() and (a[2] == full[2][100 * k:200 * k]).all()
    assert (a[3] == full[3][100:200]).all()
    assert (G.sdd_rows(n, k, 1)[2] != G.sdd_rows(n, k, 2)[2]).any()


def test_gen1000_is_dominant_and_seeded():
    rp, ci, va, b = G.gen1000_dense(size=60, seed=42)
    rp2, ci2, va2, _ = G.gen1000_dense(size=60, seed=42)
    assert (ci == ci2).all() and (va == va2).all()
    n = 60
    for i in range(n):
        cols, vals = ci[rp[i]:rp[i + 1]], va[rp[i]:rp[i + 1]]
        d = vals[cols == i][0]
        off = np.abs(vals[cols != i]).sum()
        assert abs(d - (2.0 * off + 1.0)) < 1e-12      # mcp/tools/matrix.ts:297-322


def test_compute_entry_points_fail_loudly_without_gpu():
    lib = L.load()
    n = ctypes.c_int(0)
    lib.sl_device_count(ctypes.byref(n))
    if n.value > 0:
        pytest.skip("a GPU is present")
    from sublinear_time_solver_amd import SparseMatrix, SolverError
    with pytest.raises(SolverError) as e:
        SparseMatrix.from_triplets([(0, 0, 1.0)], 1, 1)
    assert e.value.kind == "DeviceError" and "no CPU fallback" in str(e.value)


def test_balanced_row_bounds_match_the_python_partition():
    """sl_balanced_row_bounds (host arithmetic) == distributed.nnz_balanced_bounds on ragged row_ptrs"""
    from sublinear_time_solver_amd import distributed as D
    lib = L.load()
    rng = np.random.default_rng(3)
    for n, world in ((5, 2), (1000, 8), (17, 4), (1, 3), (6000, 16)):
        cnt = rng.integers(0, 40, size=n)
        cnt[rng.integers(0, n)] = 5000                                     # a hub row
        rp = np.zeros(n + 1, dtype=np.uint32)
        rp[1:] = np.cumsum(cnt)
        out = np.zeros(world + 1, dtype=np.uint64)
        assert lib.sl_balanced_row_bounds(n, L.ptr(rp), world, L.ptr(out)) == 0
        assert out.tolist() == D.nnz_balanced_bounds(rp, world)
    assert lib.sl_balanced_row_bounds(5, None, 2, L.ptr(np.zeros(3, dtype=np.uint64))) == 4


def test_communicator_fails_loudly_without_gpu_and_validates_its_arguments():
    lib = L.load()
    h = ctypes.c_void_p()
    assert lib.sl_comm_create(2, 2, b"x", ctypes.byref(h)) == 4 and not h.value            # rank out of range: InvalidInput
    assert lib.sl_comm_create(0, 17, b"x", ctypes.byref(h)) == 4                            # more ranks than one node holds
    assert lib.sl_comm_create(0, 1, b"a/b", ctypes.byref(h)) == 4                           # the rendezvous is a name, not a path
    n = ctypes.c_int(0)
    lib.sl_device_count(ctypes.byref(n))
    if n.value == 0:
        assert lib.sl_comm_create(0, 1, b"cpu_only_test", ctypes.byref(h)) == 11 and not h.value   # DeviceError, no CPU fallback
        assert b"no CPU fallback" in lib.sl_last_error_message()


def test_triplet_validation_happens_before_any_device_work():
    from sublinear_time_solver_amd import SparseMatrix, SolverError
    with pytest.raises(SolverError) as e:
        SparseMatrix.from_triplets([(0, 0, 1.0), (5, 0, 1.0)], 2, 2)
    assert e.value.kind == "IndexOutOfBounds"          # matrix/mod.rs:167-173
    with pytest.raises(SolverError) as e:
        SparseMatrix.from_triplets([(0, 0, float("inf"))], 2, 2)
    assert e.value.kind == "InvalidInput"              # matrix/mod.rs:181-186


def test_product_package_never_imports_the_oracle():
    for f in (ROOT / "sublinear_time_solver_amd").rglob("*"):
        if f.suffix in (".py", ".hip", ".hpp", ".cpp", ".h") and f.is_file():
            t = f.read_text()
            assert "oracle" not in t.lower() or f.name == "README.md", f"{f} mentions the oracle"


def test_index_width_limits_are_enforced_before_any_device_work():
    """IndexType = u32 (types.rs:22): dimensions / nnz beyond 2^32-1 are InvalidInput, not a crash."""
    import ctypes as C
    lib = L.load()
    rp = np.zeros(4, dtype=np.uint32)
    ci = np.zeros(1, dtype=np.uint32)
    va = np.zeros(1, dtype=np.float64)
    h = L.vp()
    for rows, cols, nnz in ((3, 3, 2 ** 32), (2 ** 32 + 1, 3, 1), (3, 2 ** 33, 1)):
        st = lib.sl_matrix_create_csr(rows, cols, nnz, L.ptr(rp), L.ptr(ci), L.ptr(va), L.SL_MEM_HOST, 0, 0, C.byref(h))
        assert st == 4 and b"u32" in lib.sl_last_error_message()
    st = lib.sl_matrix_create_csr(3, 3, 1, None, L.ptr(ci), L.ptr(va), L.SL_MEM_HOST, 0, 0, C.byref(h))
    assert st == 4


def test_header_is_plain_c99_and_the_library_links_from_c(c_smoke_exe):
    """include/sublinear_hip.h must be consumable by a C compiler (cgo / bindgen / ctypes generators): strict C99 build of
    tests/c/abi_smoke.c; without a GPU the program checks the loud SL_DEVICE_ERROR"""
    import subprocess
    r = subprocess.run([str(c_smoke_exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr


def test_javascript_surface_host_logic(node_surface_script):
    """bindings/node: the reference's shipped SublinearSolver surface in JavaScript over the N-API addon — config and
    matrix validation, analyzeMatrix, error codes, and the loud DeviceError without a GPU (tests/js/surface_test.js)"""
    import subprocess
    r = subprocess.run(["node", str(node_surface_script)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "host logic ok" in r.stdout, r.stdout + r.stderr


def test_every_abi_entry_is_documented_with_its_reference_counterpart():
    """INTEGRATION.md lists, for every entry point of include/sublinear_hip.h, the reference interface it replaces"""
    root = Path(__file__).resolve().parent.parent
    header = (root / "include" / "sublinear_hip.h").read_text()
    doc = (root / "INTEGRATION.md").read_text()
    names = set(re.findall(r"\b(sl_[a-z0-9_]+)\s*\(", header))
    missing = sorted(n for n in names if n not in doc)
    assert not missing, missing


def test_layout_stress_tool_builds_as_strict_c99_against_the_header(tmp_path):
    """tools/layout_stress.c (the layout build alone under many-process contention) is a C99 program over the ABI: it must keep building"""
    import subprocess
    root = Path(__file__).resolve().parent.parent
    pkg = root / "sublinear_time_solver_amd"
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-O2", f"-I{root / 'include'}", str(root / "tools" / "layout_stress.c"), "-o",
                        str(tmp_path / "layout_stress"), f"-L{pkg}", "-lsublinear_hip", "-lm", f"-Wl,-rpath,{pkg}"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(tmp_path / "layout_stress")], capture_output=True, text=True)       # no arguments: the usage line, exit status 2
    assert r.returncode == 2 and "usage" in r.stderr
