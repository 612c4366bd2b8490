"""Golden vectors of the reference's TypeScript forward push AS WRITTEN: tests/golden/reference_ts_push.npz  (G12).

  G12  SublinearSolver.solveForwardPush (src/core/solver.ts:437-522) — Gauss-Southwell: solution, iterations (= pushes), residual,
       on seeded diagonally dominant systems (S-DD rows, a tridiagonal, a power-law-ish graph system) at several epsilons, and one case
       that runs into maxIterations (the reference throws CONVERGENCE_FAILED there: recorded as converged = 0 with the iterate it held).

The reference's own tests hold ONE case of this path (G6: tests/mcp/mcp-tool-tests.js:27-52).  Like make_golden_walk.py this script READS the
method bodies from /root/reference when it runs (solveForwardPush, MatrixOperations.getEntry / getDiagonal, VectorOperations.zeros / norm2),
removes the TypeScript annotations inside them and lets node evaluate them in a scratch directory; nothing of the reference's text enters
the repository, the fixture holds inputs and the numbers the reference's code produced.

    python tests/golden/make_golden_ts_push.py
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
from make_golden_walk import REF, method_body, strip_types      # noqa: E402  (the same extraction rules)


def build_runner() -> str:
    solver = (REF / "src" / "core" / "solver.ts").read_text()
    utils = (REF / "src" / "core" / "utils.ts").read_text()
    matrix = (REF / "src" / "core" / "matrix.ts").read_text()
    push = strip_types(method_body(solver, r"private async solveForwardPush\s*"))
    push = push.replace("timeoutController?.checkTimeout();", "").replace("VectorOperations.", "")
    get_entry = strip_types(method_body(matrix, r"static getEntry\s*"))
    get_diag = strip_types(method_body(matrix, r"static getDiagonal\s*"))
    zeros = strip_types(method_body(utils, r"static zeros\s*"))
    norm2 = strip_types(method_body(utils, r"static norm2\s*"))
    return f"""
class SolverError extends Error {{ constructor(m, c, d) {{ super(m); this.code = c; this.details = d; }} }}
const ErrorCodes = new Proxy({{}}, {{ get: (_, k) => k }});
function getEntry(matrix, row, col) {{ {get_entry} }}
function getDiagonal(matrix, i) {{ {get_diag} }}
function zeros(length) {{ {zeros} }}
function norm2(vector) {{ {norm2} }}
const performanceMonitor = {{ getElapsedTime: () => 0, getMemoryIncrease: () => 0 }};
let config = null, lastState = null;
async function solveForwardPush(matrix, vector, progressCallback) {{ {push} }}
(async () => {{
  const cases = require('./cases.json');
  const out = [];
  for (const c of cases) {{
    const matrix = {{ rows: c.n, cols: c.n, format: 'coo', values: c.values, rowIndices: c.rows, colIndices: c.cols }};
    config = {{ epsilon: c.epsilon, maxIterations: c.maxIterations }};
    try {{
      const r = await solveForwardPush(matrix, c.b, undefined);
      out.push({{ name: c.name, converged: 1, solution: r.solution, iterations: r.iterations, residual: r.residual }});
    }} catch (e) {{
      if (e.code !== 'CONVERGENCE_FAILED') throw e;
      out.push({{ name: c.name, converged: 0, finalResidual: e.details.finalResidual }});
    }}
  }}
  process.stdout.write(JSON.stringify(out));
}})();
"""


def systems():
    from sublinear_time_solver_amd import generators as G
    from oracle import oracle as O
    rng = np.random.default_rng(606)
    cases = []

    def add(name, rp, ci, va, b, epsilon, max_iterations):
        rows = np.repeat(np.arange(rp.size - 1), np.diff(rp))
        cases.append(dict(name=name, n=int(rp.size - 1), rows=rows.tolist(), cols=ci.tolist(), values=va.tolist(), b=np.asarray(b, dtype=np.float64).tolist(),
                          epsilon=epsilon, maxIterations=max_iterations))

    rp, ci, va, b = G.sdd_rows(60, 6, seed=2)
    add("sdd60_eps1e-6", rp, ci, va, b, 1e-6, 100000)
    add("sdd60_randb_eps1e-10", rp, ci, va, rng.standard_normal(60) * 5.0, 1e-10, 100000)
    rp, ci, va, b = G.sdd_rows(200, 9, seed=7, half_bandwidth=20)
    add("band200_eps1e-8", rp, ci, va, rng.standard_normal(200), 1e-8, 100000)
    add("band200_cutoff_at_150_pushes", rp, ci, va, rng.standard_normal(200), 1e-12, 150)
    n = 30
    tr, tc, tv = [], [], []
    for i in range(n):
        for j, v in ((i - 1, -1.0), (i, 4.0), (i + 1, -2.5)):
            if 0 <= j < n:
                tr.append(i), tc.append(j), tv.append(v)
    rp, ci, va = O.csr_from_triplets(tr, tc, tv, n, n)
    e = np.zeros(n); e[0] = 1.0; e[n - 1] = -3.0
    add("tridiag30_asym_eps1e-9", rp, ci, va, e, 1e-9, 100000)
    # column-dominant PageRank-type system I - 0.85 P^T of a small digraph with a hub: ties of |r_i| (equal residuals) test "the FIRST largest"
    n = 40
    A = (rng.random((n, n)) < 0.12).astype(np.float64)
    A[:, 3] = 1.0
    np.fill_diagonal(A, 0.0)
    out_deg = A.sum(axis=1)
    tr, tc, tv = [], [], []
    for i in range(n):
        tr.append(i), tc.append(i), tv.append(1.0)
        for j in range(n):
            if A[j, i] and out_deg[j] > 0:
                tr.append(i), tc.append(j), tv.append(-0.85 / out_deg[j])
    rp, ci, va = O.csr_from_triplets(tr, tc, tv, n, n)
    add("pagerank40_uniform_b_eps1e-7", rp, ci, va, np.full(n, 0.15 / n), 1e-7, 100000)
    return cases


def main():
    cases = systems()
    with tempfile.TemporaryDirectory(prefix="golden_ts_push_") as d:
        scratch = Path(d)
        (scratch / "cases.json").write_text(json.dumps(cases))
        (scratch / "run.js").write_text(build_runner())
        p = subprocess.run(["node", "run.js"], cwd=scratch, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-3000:]
        res = json.loads(p.stdout)
    out = {"names": np.array([c["name"] for c in cases])}
    for c, r in zip(cases, res):
        k = c["name"]
        out[k + "/rows"] = np.array(c["rows"], dtype=np.uint32)
        out[k + "/cols"] = np.array(c["cols"], dtype=np.uint32)
        out[k + "/values"] = np.array(c["values"], dtype=np.float64)
        out[k + "/b"] = np.array(c["b"], dtype=np.float64)
        out[k + "/params"] = np.array([c["n"], c["maxIterations"], r["converged"], r.get("iterations", c["maxIterations"])], dtype=np.int64)
        out[k + "/epsilon_residual"] = np.array([c["epsilon"], r["residual"] if r["converged"] else r["finalResidual"]], dtype=np.float64)
        if r["converged"]:
            out[k + "/solution"] = np.array(r["solution"], dtype=np.float64)
    # the small cases as JSON for the JavaScript surface test
    js = [{"name": c["name"], "matrix": {"rows": c["n"], "cols": c["n"], "format": "coo", "values": c["values"], "rowIndices": c["rows"], "colIndices": c["cols"]},
           "b": c["b"], "epsilon": c["epsilon"], "maxIterations": c["maxIterations"], "solution": r["solution"], "iterations": r["iterations"], "residual": r["residual"]}
          for c, r in zip(cases, res) if c["n"] <= 60 and r["converged"]]
    (ROOT / "tests" / "golden" / "reference_ts_push_js.json").write_text(json.dumps(js))
    path = ROOT / "tests" / "golden" / "reference_ts_push.npz"
    np.savez_compressed(path, **out)
    print(path, {str(k): (int(out[str(k) + "/params"][3]), int(out[str(k) + "/params"][2])) for k in out["names"]})


if __name__ == "__main__":
    main()
