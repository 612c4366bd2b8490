"""Golden vectors of the reference's TypeScript MatrixOperations.analyzeMatrix AS WRITTEN: tests/golden/reference_ts_analyze.json  (G14).

analyzeMatrix (src/core/matrix.ts:327-351) decides whether SublinearSolver.solve accepts a system at all ("Matrix is not diagonally
dominant", solver.ts:71-78) and which dominance it reports; it reads the matrix through getEntry / getDiagonal / getRowSum / getColumnSum
(:95-206), each with its own treatment of duplicated COO entries.  As in make_golden_walk.py the method bodies are READ from /root/reference
when this script runs, their TypeScript annotations removed, and evaluated by node; the fixture holds the input matrices (JSON matrix model)
and the objects the reference's code returned.

    python tests/golden/make_golden_ts_analyze.py
"""
import json
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(Path(__file__).resolve().parent))
from make_golden_walk import REF, method_body, strip_types      # noqa: E402


def build_runner() -> str:
    matrix = (REF / "src" / "core" / "matrix.ts").read_text()
    names = {"validateMatrix": "matrix", "getEntry": "matrix, row, col", "getDiagonal": "matrix, i", "getRowSum": "matrix, row, excludeDiagonal = false",
             "getColumnSum": "matrix, col, excludeDiagonal = false", "checkDiagonalDominance": "matrix", "isSymmetric": "matrix, tolerance = 1e-10",
             "calculateSparsity": "matrix", "analyzeMatrix": "matrix"}
    import re
    fns = []
    for name, params in names.items():
        body = method_body(matrix, rf"static {name}\s*")
        body = re.sub(r"\b(const|let)\s+(\w+)\s*:\s*[^=;\n]+=", r"\1 \2 =", body)      # const x: T = ...   (receivers stay: static methods of one class)
        body = re.sub(r"\s+as\s+[A-Z]\w*", "", body)
        fns.append(f"  static {name}({params}) {{ {body} }}")
    return """
class SolverError extends Error { constructor(m, c, d) { super(m); this.code = c; this.details = d; } }
const ErrorCodes = new Proxy({}, { get: (_, k) => k });
class MatrixOperations {
""" + "\n".join(fns) + """
}
const analyzeMatrix = (m) => MatrixOperations.analyzeMatrix(m);
const cases = require('./cases.json');
const out = cases.map(c => { if (c.invalid) { try { analyzeMatrix(c.matrix); return { name: c.name, error: null }; } catch (e) { return { name: c.name, error: { message: e.message, code: e.code } }; } }
  const a = analyzeMatrix(c.matrix); if (a.dominanceStrength === Infinity) a.dominanceStrength = 'Infinity'; return { name: c.name, analysis: a }; });
process.stdout.write(JSON.stringify(out));
"""


def cases():
    from sublinear_time_solver_amd import generators as G
    rng = np.random.default_rng(1414)
    out = []

    def coo(name, n, r, c, v, cols=None):
        out.append({"name": name, "matrix": {"rows": n, "cols": cols or n, "format": "coo", "values": [float(x) for x in v], "rowIndices": [int(x) for x in r],
                                             "colIndices": [int(x) for x in c]}})

    def dense(name, a):
        a = np.asarray(a, dtype=np.float64)
        out.append({"name": name, "matrix": {"rows": a.shape[0], "cols": a.shape[1], "format": "dense", "data": a.tolist()}})

    rp, ci, va, _ = G.sdd_rows(50, 6, seed=4)
    rows = np.repeat(np.arange(50), np.diff(rp))
    coo("sdd50_row_dominant_coo", 50, rows, ci, va)
    perm = rng.permutation(va.size)
    coo("sdd50_shuffled_storage_order", 50, rows[perm], ci[perm], va[perm])            # the sums follow the storage order
    coo("sdd50_transposed_column_dominant", 50, ci, rows, va)
    a = rng.standard_normal((12, 12)); a += np.diag(np.abs(a).sum(axis=1) * 1.01)
    dense("dense12_row_dominant", a)
    dense("dense12_symmetric", (a + a.T) / 2 + np.eye(12) * 20)
    b = a.copy(); b[3, 3] = 0.0
    dense("dense12_zero_diagonal_at_3", b)
    b = a.copy(); b[5, 5] = 0.1
    dense("dense12_not_dominant", b)
    dense("dense3x5_not_square", rng.standard_normal((3, 5)))
    # duplicated entries: the diagonal is read by first match, the sums add every duplicate
    coo("dup_diagonal_first_match_small_then_large", 3, [0, 0, 0, 1, 1, 2, 2], [0, 0, 1, 1, 0, 2, 1], [0.5, 9.0, 1.0, 4.0, 1.0, 3.0, 1.0])
    coo("dup_offdiagonal_counts_twice", 3, [0, 0, 0, 1, 2], [0, 1, 1, 1, 2], [2.5, 1.0, 1.0, 4.0, 3.0])
    coo("stored_zero_diagonal", 3, [0, 1, 1, 2], [0, 1, 1, 2], [2.0, 0.0, 5.0, 3.0])
    coo("exactly_on_the_boundary", 2, [0, 0, 1, 1], [0, 1, 0, 1], [1.0, -1.0, 0.25, 0.5])
    coo("symmetric_coo_with_duplicates", 3, [0, 1, 0, 1, 2, 0, 1], [0, 1, 1, 0, 2, 1, 0], [5.0, 6.0, 1.0, 1.0, 7.0, 0.5, 0.25])
    # column dominant only: the PageRank system I - 0.85 P^T of a digraph with a hub (the hub's ROW collects 0.85 / out_j from everybody)
    n = 20
    A = (rng.random((n, n)) < 0.15).astype(np.float64)
    A[:, 2] = 1.0
    np.fill_diagonal(A, 0.0)
    arp = np.concatenate([[0], np.cumsum((A != 0).sum(axis=1))]).astype(np.uint32)
    rr, cc = np.nonzero(A)
    srp, sci, sva, _ = G.pagerank_system(n, arp, cc.astype(np.uint32), A[rr, cc], 0.85)
    coo("pagerank20_column_dominant_only", n, np.repeat(np.arange(n), np.diff(srp.astype(np.int64))), sci, sva)
    # validateMatrix (core/matrix.ts:11-55): which error a malformed matrix draws, message and code — checked entry after entry, row before column
    def invalid(name, matrix):
        out.append({"name": name, "invalid": True, "matrix": matrix})
    invalid("invalid_zero_rows", {"rows": 0, "cols": 3, "format": "dense", "data": []})
    invalid("invalid_dense_row_count", {"rows": 2, "cols": 2, "format": "dense", "data": [[1.0, 0.0]]})
    invalid("invalid_dense_row_length", {"rows": 2, "cols": 2, "format": "dense", "data": [[1.0, 0.0], [1.0]]})
    invalid("invalid_coo_lengths", {"rows": 2, "cols": 2, "format": "coo", "values": [1.0, 2.0], "rowIndices": [0], "colIndices": [0, 1]})
    invalid("invalid_coo_missing_array", {"rows": 2, "cols": 2, "format": "coo", "values": [1.0], "rowIndices": [0]})
    invalid("invalid_column_in_entry_0_row_in_entry_2", {"rows": 2, "cols": 2, "format": "coo", "values": [1.0, 1.0, 1.0], "rowIndices": [0, 1, 5], "colIndices": [7, 1, 0]})
    invalid("invalid_row_and_column_in_one_entry", {"rows": 2, "cols": 2, "format": "coo", "values": [1.0], "rowIndices": [-1], "colIndices": [9]})
    invalid("invalid_format", {"rows": 2, "cols": 2, "format": "csr", "data": []})
    return out


def main():
    cs = cases()
    with tempfile.TemporaryDirectory(prefix="golden_ts_an_") as d:
        scratch = Path(d)
        (scratch / "cases.json").write_text(json.dumps(cs))
        (scratch / "run.js").write_text(build_runner())
        p = subprocess.run(["node", "run.js"], cwd=scratch, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        res = json.loads(p.stdout)
    for c, r in zip(cs, res):
        if c.get("invalid"):
            c["error"] = r["error"]
        else:
            c["analysis"] = r["analysis"]
    path = ROOT / "tests" / "golden" / "reference_ts_analyze.json"
    path.write_text(json.dumps(cs))
    for c in cs:
        if c.get("invalid"):
            print(f"{c['name']:44s} {c['error']}")
            continue
        a = c["analysis"]
        print(f"{c['name']:44s} dd {a['isDiagonallyDominant']!s:5s} {a['dominanceType']:6s} strength {a['dominanceStrength']!r:24} sym {a['isSymmetric']!s:5s} sparsity {a['sparsity']}")


if __name__ == "__main__":
    main()
