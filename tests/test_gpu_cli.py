"""CLI-shaped front end end to end on the GPU: solve / analyze / pagerank / generate."""
import json
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _run(*args):
    r = subprocess.run([sys.executable, "-m", "sublinear_time_solver_amd", *args], cwd=ROOT, capture_output=True, text=True, timeout=600)
    return r


def test_generate_solve_roundtrip(gpu, tmp_path):
    a, b, x = tmp_path / "A.json", tmp_path / "b.json", tmp_path / "x.json"
    r = _run("generate", "-t", "diagonally-dominant", "-s", "120", "-o", str(a))
    assert r.returncode == 0, r.stderr
    b.write_text(json.dumps([1.0] * 120))
    r = _run("solve", "-m", str(a), "-b", str(b), "--method", "neumann", "--epsilon", "1e-10", "-o", str(x), "--verbose")
    assert r.returncode == 0, r.stderr + r.stdout
    assert "Converged: True" in r.stdout and "Diagonally dominant: True" in r.stdout
    sol = json.loads(x.read_text())
    A = np.asarray(json.loads(a.read_text())["data"])
    assert np.linalg.norm(A @ np.asarray(sol["solution"]) - 1.0) < 1e-8 and sol["method"] == "neumann"
    r2 = _run("solve", "-m", str(a), "-b", str(b), "--method", "forward-push", "--epsilon", "1e-10", "-o", str(tmp_path / "x2.json"))
    assert r2.returncode == 0, r2.stderr
    sol2 = json.loads((tmp_path / "x2.json").read_text())
    np.testing.assert_allclose(sol2["solution"], sol["solution"], atol=1e-8)


def test_analyze_and_pagerank(gpu, tmp_path):
    m = tmp_path / "t.mtx"
    m.write_text("%%MatrixMarket matrix coordinate real general\n3 3 7\n1 1 4\n1 2 -1\n2 1 -1\n2 2 4\n2 3 -1\n3 2 -1\n3 3 3\n")
    r = _run("analyze", "-m", str(m))
    assert r.returncode == 0, r.stderr
    an = json.loads(r.stdout)
    assert an["isDiagonallyDominant"] and an["dominanceType"] == "row" and an["isSymmetric"]
    g = tmp_path / "g.json"
    adj = [[0, 1, 1, 0], [1, 0, 0, 1], [0, 1, 0, 1], [1, 0, 1, 0]]
    g.write_text(json.dumps({"rows": 4, "cols": 4, "format": "dense", "data": adj}))
    r = _run("pagerank", "-g", str(g), "--damping", "0.85", "--epsilon", "1e-12")
    assert r.returncode == 0, r.stderr
    pr = json.loads(r.stdout)
    A = np.asarray(adj, dtype=float)
    P = A / A.sum(axis=1, keepdims=True)
    x = np.linalg.solve(np.eye(4) - 0.85 * P.T, np.full(4, 0.15 / 4))
    assert pr["converged"] and abs(pr["totalScore"] - 1.0) < 1e-9
    assert pr["topNodes"][0]["node"] == int(np.argmax(x)) and abs(pr["topNodes"][0]["score"] - x.max()) < 1e-9


def test_cli_reports_solver_errors(gpu, tmp_path):
    a, b = tmp_path / "A.json", tmp_path / "b.json"
    a.write_text(json.dumps({"rows": 2, "cols": 2, "format": "dense", "data": [[1, 5], [5, 1]]}))
    b.write_text("[1, 1]")
    r = _run("solve", "-m", str(a), "-b", str(b))
    assert r.returncode == 1 and "MatrixNotDiagonallyDominant" in r.stderr and "Warning: Matrix is not diagonally dominant" in r.stderr


def test_c_program_solves_through_the_abi(gpu, c_smoke_exe):
    """a C99 program (gcc, no HIP headers) drives create / solve / estimate_entry through include/sublinear_hip.h"""
    import subprocess
    r = subprocess.run([str(c_smoke_exe), "gpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok:"), r.stdout + r.stderr


def test_javascript_surface_on_gpu(gpu, node_surface_script):
    """the same surface solving on the GPU: solve (dense / coo / CLI-nested coo, all methods), estimateEntry (push and
    seeded random walk), computePageRank vs power iteration (tests/js/surface_test.js gpu)"""
    import subprocess
    r = subprocess.run(["node", str(node_surface_script), "gpu"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "gpu surface ok" in r.stdout, r.stdout + r.stderr
