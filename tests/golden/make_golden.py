#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the RUNNABLE members of the reference:

  G1  Python `IterativeSolvers.jacobi` (scripts/linear_systems/iterative_solvers.py:17-104) on the
      reference's own fixtures scripts/linear_systems/test_matrices/n_*/*.json: iterates after
      k = 1, 2, 5, 10 sweeps and the converged solution (tol 1e-12).  Jacobi from x0 = 0 after K
      sweeps == the Neumann partial sum of K terms (neumann.rs:252-299 with a zero initial guess).
  G2  JS `JSSolver` Jacobi (src/solver.js:275-358) final x / iterations under its relative rule.

Runs ONLY in the build container (needs /root/reference; imports the Python reference with
sys.dont_write_bytecode, copies the JS files to a scratch dir outside the reference tree and
deletes them).  Only derived data — inputs as CSR, expected outputs — is written here.
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent
FIXTURES = ["n_50/dd_asymmetric", "n_100/banded", "n_100/tridiagonal", "n_100/dd_asymmetric", "n_200/dd_asymmetric"]
# the reference's "sparse_dd" fixtures are NOT row dominant (Jacobi diverges on n_100): kept as the negative case
NEGATIVE = ["n_100/sparse_dd"]
RHS = ["ones", "random", "smooth"]
KS = [1, 2, 5, 10]

sys.dont_write_bytecode = True
sys.path.insert(0, str(REF / "scripts" / "linear_systems"))
from iterative_solvers import IterativeSolvers  # noqa: E402


def to_csr(A):
    rp = [0]
    ci, va = [], []
    for i in range(A.shape[0]):
        nz = np.nonzero(A[i])[0]
        ci.extend(nz.tolist())
        va.extend(A[i, nz].tolist())
        rp.append(len(ci))
    return np.asarray(rp, np.uint32), np.asarray(ci, np.uint32), np.asarray(va, np.float64)


def js_jacobi(cases):
    """cases: list of (A dense list, b list, tol, maxit) -> list of dict(values, iterations, residual, converged)"""
    scratch = Path(tempfile.mkdtemp(prefix="slgold_"))
    try:
        (scratch / "src" / "convergence").mkdir(parents=True)
        (scratch / "src" / "utils").mkdir(parents=True)
        shutil.copy(REF / "src" / "solver.js", scratch / "src" / "solver.js")
        for f in (REF / "src" / "convergence").glob("*.js"):
            shutil.copy(f, scratch / "src" / "convergence" / f.name)
        shutil.copy(REF / "src" / "utils" / "matrix-utils.js", scratch / "src" / "utils" / "matrix-utils.js")
        (scratch / "cases.json").write_text(json.dumps(cases))
        (scratch / "run.js").write_text("""
const { createSolver } = require('./src/solver.js');
const cases = require('./cases.json');
(async () => {
  const out = [];
  for (const c of cases) {
    const solver = await createSolver({ matrix: { rows: c.n, cols: c.n, format: 'dense', data: c.A },
                                  method: 'jacobi', tolerance: c.tol, maxIterations: c.maxit, verbose: false });
    const r = await solver.solve(c.b);
    out.push({ values: Array.from(r.values), iterations: r.iterations, residual: r.residual, converged: r.converged });
  }
  process.stdout.write('@@RESULT@@' + JSON.stringify(out));
})().catch(e => { console.error(e); process.exit(1); });
""")
        p = subprocess.run(["node", "run.js"], cwd=scratch, capture_output=True, text=True, timeout=600)
        if p.returncode != 0:
            raise RuntimeError(p.stderr[-2000:])
        return json.loads(p.stdout.split("@@RESULT@@")[1])
    finally:
        shutil.rmtree(scratch, ignore_errors=True)


def main():
    solver = IterativeSolvers()
    js_cases = []
    store = {}
    index = []
    for fx in FIXTURES:
        d = json.load(open(REF / "scripts" / "linear_systems" / "test_matrices" / f"{fx}.json"))
        A = np.asarray(d["matrix"], dtype=np.float64)
        n = A.shape[0]
        rp, ci, va = to_csr(A)
        key = fx.replace("/", "_")
        store[f"{key}__row_ptr"], store[f"{key}__col_idx"], store[f"{key}__values"] = rp, ci, va
        for rn in RHS:
            b = np.asarray(d["rhs_vectors"][rn], dtype=np.float64)
            snaps = {}

            def cb(k, x, res, snaps=snaps):
                if (k + 1) in KS:
                    snaps[k + 1] = x.copy()

            r = solver.jacobi(A, b, max_iter=2000, tol=1e-12, callback=cb)
            assert r["success"], (fx, rn)
            ck = f"{key}__{rn}"
            store[f"{ck}__b"] = b
            for k in KS:
                store[f"{ck}__x_k{k}"] = snaps[k]
            store[f"{ck}__x_final"] = np.asarray(r["solution"])
            store[f"{ck}__py_iterations"] = np.asarray([r["iterations"]])
            index.append(ck)
            js_cases.append({"n": n, "A": A.tolist(), "b": b.tolist(), "tol": 1e-10, "maxit": 2000})
    for fx in NEGATIVE:
        d = json.load(open(REF / "scripts" / "linear_systems" / "test_matrices" / f"{fx}.json"))
        A = np.asarray(d["matrix"], dtype=np.float64)
        rp, ci, va = to_csr(A)
        key = "neg_" + fx.replace("/", "_")
        store[f"{key}__row_ptr"], store[f"{key}__col_idx"], store[f"{key}__values"] = rp, ci, va
    js = js_jacobi(js_cases)
    for ck, r in zip(index, js):
        store[f"{ck}__js_x"] = np.asarray(r["values"], dtype=np.float64)
        store[f"{ck}__js_iterations"] = np.asarray([r["iterations"]])
        store[f"{ck}__js_residual"] = np.asarray([r["residual"]])
        assert r["converged"], ck
    store["__cases"] = np.asarray(index)
    np.savez_compressed(OUT / "reference_jacobi.npz", **store)
    print("wrote", OUT / "reference_jacobi.npz", os.path.getsize(OUT / "reference_jacobi.npz"), "bytes;", len(index), "cases")


if __name__ == "__main__":
    main()
