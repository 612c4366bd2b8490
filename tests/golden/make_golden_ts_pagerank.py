"""Golden vectors of the reference's TypeScript computePageRank AS WRITTEN: tests/golden/reference_ts_pagerank.npz  (G13).

  G13  SublinearSolver.computePageRank (src/core/solver.ts:664-722) with method 'forward-push': the system matrix it assembles
       (out-degrees as left-to-right row sums, S[i][j] = [i == j] - damping * (adj[j][i] / out_j), dangling columns untouched), the
       right-hand side, and the solution / iteration count its own solveForwardPush (:437-522) returns for it.

G9 pins the PageRank path against the reference's runnable Python power iteration to 1e-13; this pins the ASSEMBLY ARITHMETIC and the solve
bit for bit.  Like make_golden_walk.py, the method bodies (computePageRank, solveForwardPush, MatrixOperations.getEntry / getDiagonal,
VectorOperations.zeros / ones / scale / norm2) are READ from /root/reference when this script runs, their TypeScript annotations removed, and
evaluated by node in a scratch directory.  `new SublinearSolver(cfg).solve(matrix, rhs)` — validation, analysis and the dispatch on
cfg.method — is stood in for by a direct call of the extracted solveForwardPush with cfg's epsilon / maxIterations (the dispatch target of
'forward-push', solver.ts:92-95).  Nothing of the reference's text enters the repository.

    python tests/golden/make_golden_ts_pagerank.py
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
    solver = (REF / "src" / "core" / "solver.ts").read_text()
    utils = (REF / "src" / "core" / "utils.ts").read_text()
    matrix = (REF / "src" / "core" / "matrix.ts").read_text()
    push = strip_types(method_body(solver, r"private async solveForwardPush\s*")).replace("timeoutController?.checkTimeout();", "").replace("VectorOperations.", "")
    pr = strip_types(method_body(solver, r"async computePageRank\s*")).replace("VectorOperations.", "")
    pr = pr.replace("validateMatrix(adjacency);", "").replace("ValidationUtils.validateRange(config.damping, 0, 1, 'damping');", "")
    pr = pr.replace("ValidationUtils.validatePositiveNumber(config.epsilon, 'epsilon');", "")
    # `this.config.method` / `this.config.timeout` lost their receiver in strip_types: the outer solver's config is `outer`
    pr = pr.replace("method: config.method,", "method: outer.method,").replace("timeout: config.timeout", "timeout: outer.timeout")
    pr = pr.replace("const solver = new SublinearSolver(solverConfig);", "const solver = { solve: (m, v) => { LAST = { matrix: m, rhs: v }; pushConfig = solverConfig; return solveForwardPush(m, v, undefined); } };")
    get_entry = strip_types(method_body(matrix, r"static getEntry\s*"))
    get_diag = strip_types(method_body(matrix, r"static getDiagonal\s*"))
    fn = {k: strip_types(method_body(utils, rf"static {k}\s*")) for k in ("zeros", "ones", "scale", "norm2")}
    push = push.replace("config.maxIterations", "pushConfig.maxIterations").replace("config.epsilon", "pushConfig.epsilon")
    return f"""
class SolverError extends Error {{ constructor(m, c, d) {{ super(m); this.code = c; this.details = d; }} }}
const ErrorCodes = new Proxy({{}}, {{ get: (_, k) => k }});
function getEntry(matrix, row, col) {{ {get_entry} }}
function getDiagonal(matrix, i) {{ {get_diag} }}
function zeros(length) {{ {fn['zeros']} }}
function ones(length) {{ {fn['ones']} }}
function scale(vector, scalar) {{ {fn['scale']} }}
function norm2(vector) {{ {fn['norm2']} }}
const performanceMonitor = {{ getElapsedTime: () => 0, getMemoryIncrease: () => 0 }};
let pushConfig = null, LAST = null, ITER = 0;
const outer = {{ method: 'forward-push', timeout: undefined }};
async function solveForwardPush(matrix, vector, progressCallback) {{ {push} }}
async function computePageRank(adjacency, config) {{ {pr} }}
(async () => {{
  const cases = require('./cases.json');
  const out = [];
  for (const c of cases) {{
    const adjacency = {{ rows: c.n, cols: c.n, format: 'coo', values: c.values, rowIndices: c.rows, colIndices: c.cols }};
    const cfg = {{ damping: c.damping, epsilon: c.epsilon, maxIterations: c.maxIterations }};
    if (c.personalized) cfg.personalized = c.personalized;
    const solution = await computePageRank(adjacency, cfg);
    // iterations: re-run the push on the system the reference assembled (same inputs, same code: the count it reported internally)
    const again = await solveForwardPush(LAST.matrix, LAST.rhs, undefined);
    out.push({{ name: c.name, system: LAST.matrix.data, rhs: LAST.rhs, solution, iterations: again.iterations, residual: again.residual }});
  }}
  process.stdout.write(JSON.stringify(out));
}})();
"""


def digraph(rng, n, density, hub=None, self_loops=(), dangling=(), weighted=False):
    A = (rng.random((n, n)) < density).astype(np.float64)
    np.fill_diagonal(A, 0.0)
    if hub is not None:
        A[:, hub] = 1.0
        A[hub, hub] = 0.0
    for i in self_loops:
        A[i, i] = 1.0
    if weighted:
        A *= np.round(rng.uniform(0.25, 4.0, size=A.shape), 2)
    for i in dangling:
        A[i, :] = 0.0
    rr, cc = np.nonzero(A)
    return rr, cc, A[rr, cc]


def main():
    rng = np.random.default_rng(1313)
    cases = []
    for name, n, kw, damping, eps, pers in (
            ("unweighted30_d0.85", 30, dict(density=0.15, hub=3, dangling=(7, 21)), 0.85, 1e-8, False),
            ("weighted40_selfloops_d0.9", 40, dict(density=0.12, hub=0, self_loops=(5, 6, 17), dangling=(11,), weighted=True), 0.9, 1e-9, False),
            ("weighted25_personalized_d0.7", 25, dict(density=0.2, self_loops=(2,), weighted=True), 0.7, 1e-10, True)):
        rr, cc, vv = digraph(rng, n, **kw)
        c = dict(name=name, n=n, rows=rr.tolist(), cols=cc.tolist(), values=vv.tolist(), damping=damping, epsilon=eps, maxIterations=200000)
        if pers:
            p = rng.random(n)
            c["personalized"] = (p / p.sum() * (1 - damping)).tolist()
        cases.append(c)
    with tempfile.TemporaryDirectory(prefix="golden_ts_pr_") as d:
        scratch = Path(d)
        (scratch / "cases.json").write_text(json.dumps(cases))
        (scratch / "run.js").write_text(build_runner())
        p = subprocess.run(["node", "run.js"], cwd=scratch, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-3000:]
        res = json.loads(p.stdout)
    out = {"names": np.array([c["name"] for c in cases])}
    for c, r in zip(cases, res):
        k = c["name"]
        out[k + "/adj_rows"] = np.array(c["rows"], dtype=np.uint32)
        out[k + "/adj_cols"] = np.array(c["cols"], dtype=np.uint32)
        out[k + "/adj_values"] = np.array(c["values"], dtype=np.float64)
        out[k + "/params"] = np.array([c["n"], c["maxIterations"], r["iterations"], 1 if "personalized" in c else 0], dtype=np.int64)
        out[k + "/damping_epsilon_residual"] = np.array([c["damping"], c["epsilon"], r["residual"]], dtype=np.float64)
        out[k + "/system"] = np.array(r["system"], dtype=np.float64)               # the dense n x n table the reference assembled
        out[k + "/rhs"] = np.array(r["rhs"], dtype=np.float64)
        out[k + "/solution"] = np.array(r["solution"], dtype=np.float64)
    # the same cases as JSON for the JavaScript surface test (node reads no .npz): the system as sorted (row, col, value) triplets
    js = []
    for c, r in zip(cases, res):
        S = np.array(r["system"], dtype=np.float64)
        rr, cc = np.nonzero(S)
        js.append({"name": c["name"], "adjacency": {"rows": c["n"], "cols": c["n"], "format": "coo", "values": c["values"], "rowIndices": c["rows"], "colIndices": c["cols"]},
                   "damping": c["damping"], "epsilon": c["epsilon"], "maxIterations": c["maxIterations"], "personalized": c.get("personalized"),
                   "system": {"rows": rr.tolist(), "cols": cc.tolist(), "values": S[rr, cc].tolist()}, "rhs": r["rhs"], "solution": r["solution"], "iterations": r["iterations"]})
    (ROOT / "tests" / "golden" / "reference_ts_pagerank_js.json").write_text(json.dumps(js))
    path = ROOT / "tests" / "golden" / "reference_ts_pagerank.npz"
    np.savez_compressed(path, **out)
    print(path, {str(k): int(out[str(k) + "/params"][2]) for k in out["names"]})


if __name__ == "__main__":
    main()
