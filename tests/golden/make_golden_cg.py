#!/usr/bin/env python3
"""G8 — golden vectors for the CG path from the RUNNABLE JavaScript twin of FastConjugateGradient
(js/fast-solver.js:108-228, the same algorithm as src/fast_solver.rs:126-178 / src/optimized_solver.rs:182-295) on the
reference's own symmetric positive definite fixtures scripts/linear_systems/test_matrices/n_*/{dd_symmetric, laplacian_1d,
laplacian_2d, spd_well_conditioned, spd_ill_conditioned}.json with their three right-hand sides.

Runs ONLY in the build container (needs /root/reference and node): the JS file is copied to a scratch directory outside
the reference tree as an ES module, driven, and deleted.  Only derived data — inputs as CSR, the returned solution and the
number of matrix-vector products the solver made — is written to tests/golden/reference_cg.npz."""
import json
import os
import shutil
import subprocess
import tempfile
from pathlib import Path

import numpy as np

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent
FIXTURES = ["n_50/dd_symmetric", "n_50/laplacian_1d", "n_50/spd_well_conditioned", "n_50/spd_ill_conditioned",
            "n_100/dd_symmetric", "n_100/laplacian_1d", "n_100/laplacian_2d", "n_100/spd_well_conditioned", "n_100/spd_ill_conditioned",
            "n_200/dd_symmetric", "n_200/laplacian_1d", "n_200/spd_well_conditioned", "n_500/laplacian_1d"]
RHS = ["ones", "random", "smooth"]
TOL, MAXIT = 1e-10, 1000

DRIVER = """
import { FastCSRMatrix, FastConjugateGradient } from './fast-solver.mjs';
import { readFileSync } from 'fs';
const cases = JSON.parse(readFileSync('./cases.json', 'utf8'));
const out = [];
for (const c of cases) {
  const m = new FastCSRMatrix(c.values, c.col_idx, c.row_ptr, c.n, c.n);
  let products = 0;
  const mv = m.multiplyVector.bind(m);
  m.multiplyVector = (x, y) => { products++; return mv(x, y); };
  const x = new FastConjugateGradient(c.maxit, c.tol).solve(m, c.b);
  out.push({ x: Array.from(x), products });
}
process.stdout.write('@@RESULT@@' + JSON.stringify(out));
"""


def to_csr(A):
    rp, ci, va = [0], [], []
    for i in range(A.shape[0]):
        nz = np.nonzero(A[i])[0]
        ci.extend(nz.tolist())
        va.extend(A[i, nz].tolist())
        rp.append(len(ci))
    return np.asarray(rp, np.uint32), np.asarray(ci, np.uint32), np.asarray(va, np.float64)


def main():
    cases, index, store = [], [], {}
    for fx in FIXTURES:
        d = json.load(open(REF / "scripts" / "linear_systems" / "test_matrices" / f"{fx}.json"))
        A = np.asarray(d["matrix"], dtype=np.float64)
        assert np.allclose(A, A.T, rtol=1e-9, atol=1e-12), fx
        rp, ci, va = to_csr(A)
        key = fx.replace("/", "_")
        store[f"{key}__row_ptr"], store[f"{key}__col_idx"], store[f"{key}__values"] = rp, ci, va
        for rn in RHS:
            b = np.asarray(d["rhs_vectors"][rn], dtype=np.float64)
            store[f"{key}__{rn}__b"] = b
            index.append(f"{key}__{rn}")
            cases.append({"n": int(A.shape[0]), "row_ptr": rp.tolist(), "col_idx": ci.tolist(), "values": va.tolist(), "b": b.tolist(),
                          "tol": TOL, "maxit": MAXIT})
    scratch = Path(tempfile.mkdtemp(prefix="slgoldcg_"))
    try:
        shutil.copy(REF / "js" / "fast-solver.js", scratch / "fast-solver.mjs")
        (scratch / "cases.json").write_text(json.dumps(cases))
        (scratch / "run.mjs").write_text(DRIVER)
        p = subprocess.run(["node", "--experimental-modules", "run.mjs"], cwd=scratch, capture_output=True, text=True, timeout=600)
        if p.returncode != 0:
            raise RuntimeError(p.stderr[-2000:])
        res = json.loads(p.stdout.split("@@RESULT@@")[1])
    finally:
        shutil.rmtree(scratch, ignore_errors=True)
    for ck, r in zip(index, res):
        store[f"{ck}__js_x"] = np.asarray(r["x"], dtype=np.float64)
        store[f"{ck}__js_products"] = np.asarray([r["products"]])
    store["__cases"] = np.asarray(index)
    store["__tol"] = np.asarray([TOL])
    np.savez_compressed(OUT / "reference_cg.npz", **store)
    print("wrote", OUT / "reference_cg.npz", os.path.getsize(OUT / "reference_cg.npz"), "bytes;", len(index), "cases")


if __name__ == "__main__":
    main()
