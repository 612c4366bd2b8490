"""Golden vectors of the reference's random-walk code AS WRITTEN (one serial LCG stream): tests/golden/reference_walk.npz.

  G10  TS estimateEntry, method 'random-walk' (src/core/solver.ts:585-601, 630-634): per-walk estimates, mean, variance
  G11  TS solveRandomWalk (src/core/solver.ts:278-333): solution, per-coordinate variances, totalVariance, residual

Runs ONLY in the build container (needs /root/reference and node).  The reference's shipped source for this path is TypeScript and the
image has no tsc, so the functions on the path — createSeededRandom (core/utils.ts:161-168), MatrixOperations.getEntry / getDiagonal
(core/matrix.ts:95-123), createTransitionMatrix and performRandomWalk (core/solver.ts:359-432) — are READ from /root/reference when this
script runs, their bodies taken between the method's braces, the few TypeScript annotations inside them removed (`const x: T =`, `as T`,
`this.` / `MatrixOperations.` receivers), and evaluated by node in a scratch directory outside the repository.  The sample loops and the
mean / variance reductions around them are the reference's statements too (extracted by pattern from estimateEntry / solveRandomWalk).
Nothing of the reference's text is written into the repository: the fixture holds inputs (CSR arrays, b, row, epsilon, seed) and the
numbers the reference's code printed for them.

    python tests/golden/make_golden_walk.py
"""
import json
import re
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
REF = Path("/root/reference")


def method_body(text: str, header_pattern: str) -> str:
    """text between the braces of the first function whose header matches (brace counting; the header may span lines up to its `{`)"""
    m = re.search(header_pattern, text)
    assert m, header_pattern
    i = m.end()
    # the opening brace of the BODY: the first `{` after the parameter list's closing parenthesis at depth 0 that is followed by a statement,
    # skipping a `{ ... }` return-type literal (createTransitionMatrix): take the last `{` before the first line that does not belong to a type
    depth, j = 0, i
    # walk to the end of the parameter list
    while True:
        c = text[j]
        if c == "(":
            depth += 1
        elif c == ")":
            depth -= 1
            if depth == 0:
                break
        j += 1
    j += 1
    rest = text[j:]
    rt = re.match(r"\s*:\s*(?:Promise<)?\{[^{}]*\}>?\s*\{", rest)          # `): { a: T; b: U; } {` or `): Promise<{ ... }> {`
    if rt:
        start = j + rt.end()
    else:
        start = j + rest.index("{") + 1
    depth, k = 1, start
    while depth:
        if text[k] == "{":
            depth += 1
        elif text[k] == "}":
            depth -= 1
        k += 1
    return text[start:k - 1]


def strip_types(body: str) -> str:
    body = re.sub(r"\b(const|let)\s+(\w+)\s*:\s*[^=;\n]+=", r"\1 \2 =", body)      # const x: T = ...
    body = re.sub(r"\s+as\s+[A-Z]\w*", "", body)                                     # x as DenseMatrix
    body = body.replace("this.validateMatrix(matrix);", "")
    body = body.replace("this.timeoutController?.checkTimeout();", "")
    body = body.replace("MatrixOperations.", "").replace("this.", "")
    return body


def build_runner() -> str:
    solver = (REF / "src" / "core" / "solver.ts").read_text()
    utils = (REF / "src" / "core" / "utils.ts").read_text()
    matrix = (REF / "src" / "core" / "matrix.ts").read_text()
    seeded = strip_types(method_body(utils, r"export function createSeededRandom\s*"))
    get_entry = strip_types(method_body(matrix, r"static getEntry\s*"))
    get_diag = strip_types(method_body(matrix, r"static getDiagonal\s*"))
    ctm = strip_types(method_body(solver, r"private createTransitionMatrix\s*"))
    walk = strip_types(method_body(solver, r"private performRandomWalk\s*"))
    est = method_body(solver, r"async estimateEntry\s*")
    srw = method_body(solver, r"private async solveRandomWalk\s*")
    # the reductions, as the reference writes them
    mean_e = re.search(r"const mean = (estimates\.reduce\([^;]+);", est).group(1)
    var_e = re.search(r"const variance = estimates\.length > 1\s*\?\s*(estimates\.reduce\([^:]+?)\s*:\s*0;", est, re.S).group(1)
    mean_s = re.search(r"const mean = (estimates\.reduce\([^;]+);", srw).group(1)
    var_s = re.search(r"const variance = (estimates\.reduce\([^;]+);", srw).group(1)
    nsamp = re.search(r"const numSamples = ([^;]+);", est).group(1).replace("config.epsilon", "epsilon")
    nwalk = re.search(r"const numWalks = ([^;]+);", srw).group(1).replace("this.config.epsilon", "epsilon")
    norm2 = strip_types(method_body(utils, r"static norm2\s*"))
    return f"""
class SolverError extends Error {{ constructor(m, c) {{ super(m); this.code = c; }} }}
const ErrorCodes = new Proxy({{}}, {{ get: (_, k) => k }});
function createSeededRandom(seed) {{ {seeded} }}
function getEntry(matrix, row, col) {{ {get_entry} }}
function getDiagonal(matrix, i) {{ {get_diag} }}
function createTransitionMatrix(matrix) {{ {ctm} }}
function performRandomWalk(start, transitions, absorptionProbs, vector, rng) {{ {walk} }}
function norm2(vector) {{ {norm2} }}
const cases = require('./cases.json');
const out = [];
for (const c of cases) {{
  const matrix = {{ rows: c.n, cols: c.n, format: 'coo', values: c.values, rowIndices: c.rows, colIndices: c.cols }};
  const vector = c.b, epsilon = c.epsilon;
  const rng = createSeededRandom(c.seed);
  const {{ transitions, absorptionProbs }} = createTransitionMatrix(matrix);
  if (c.kind === 'estimate') {{
    const estimates = [];
    const numSamples = {nsamp};
    for (let i = 0; i < numSamples; i++) estimates.push(performRandomWalk(c.row, transitions, absorptionProbs, vector, rng));
    const mean = {mean_e};
    const variance = estimates.length > 1 ? {var_e} : 0;
    out.push({{ name: c.name, estimates, mean, variance }});
  }} else {{
    const n = c.n, solution = new Array(n).fill(0), variances = [];
    let totalVariance = 0;
    for (let i = 0; i < n; i++) {{
      const estimates = [];
      const numWalks = {nwalk};
      for (let walk = 0; walk < numWalks; walk++) estimates.push(performRandomWalk(i, transitions, absorptionProbs, vector, rng));
      const mean = {mean_s};
      const variance = {var_s};
      solution[i] = mean; totalVariance += variance; variances.push(variance);
    }}
    // residual: multiplyMatrixVector over the COO entries in stored order (core/matrix.ts:79-86), subtract, norm2
    const ax = new Array(n).fill(0);
    for (let k = 0; k < c.values.length; k++) ax[c.rows[k]] += c.values[k] * solution[c.cols[k]];
    const residual = norm2(ax.map((v, i) => v - vector[i]));
    out.push({{ name: c.name, solution, variances, totalVariance, residual }});
  }}
}}
process.stdout.write(JSON.stringify(out));
"""


def systems():
    from sublinear_time_solver_amd import generators as G
    rng = np.random.default_rng(2026)
    cases = []

    def add(kind, name, rp, ci, va, b, **kw):
        rows = np.repeat(np.arange(rp.size - 1), np.diff(rp))
        cases.append(dict(kind=kind, name=name, n=int(rp.size - 1), rows=rows.tolist(), cols=ci.tolist(), values=va.tolist(), b=b.tolist(), **kw))

    rp, ci, va, _ = G.sdd_rows(40, 5, seed=3)
    add("estimate", "sdd40_row0_eps0.1_seed42", rp, ci, va, rng.standard_normal(40) * 2.0, row=0, epsilon=0.1, seed=42)
    add("estimate", "sdd40_row17_eps0.05_seed7", rp, ci, va, rng.standard_normal(40) * 2.0, row=17, epsilon=0.05, seed=7)
    rp, ci, va, _ = G.sdd_rows(64, 6, seed=11, half_bandwidth=12)
    add("estimate", "band64_row63_eps0.08_seed123456789", rp, ci, va, rng.standard_normal(64), row=63, epsilon=0.08, seed=123456789)
    # a large diagonal (absorption probability 1 / a_ii = 0.025 per step): long walks, about 80 draws per walk
    n = 24
    tr, tc, tv = [], [], []
    for i in range(n):
        for j, v in ((i - 1, -15.0), (i, 40.0), (i + 1, 12.5)):
            if 0 <= j < n:
                tr.append(i), tc.append(j), tv.append(v)
    from oracle import oracle as O
    rp, ci, va = O.csr_from_triplets(tr, tc, tv, n, n)
    add("estimate", "long24_row12_eps0.1_seed5", rp, ci, va, np.arange(1.0, n + 1.0), row=12, epsilon=0.1, seed=5)
    rp, ci, va, _ = G.sdd_rows(12, 4, seed=5)
    add("solve", "sdd12_eps0.1_seed9", rp, ci, va, rng.standard_normal(12) * 3.0, epsilon=0.1, seed=9)
    rp, ci, va, _ = G.sdd_rows(30, 6, seed=8)
    add("solve", "sdd30_eps0.07_seed31337", rp, ci, va, rng.standard_normal(30), epsilon=0.07, seed=31337)
    return cases


def main():
    cases = systems()
    with tempfile.TemporaryDirectory(prefix="golden_walk_") as d:
        scratch = Path(d)
        (scratch / "cases.json").write_text(json.dumps(cases))
        (scratch / "run.js").write_text(build_runner())
        p = subprocess.run(["node", "run.js"], cwd=scratch, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        res = json.loads(p.stdout)
    out = {"names": np.array([c["name"] for c in cases])}
    for c, r in zip(cases, res):
        k = c["name"]
        out[k + "/rows"] = np.array(c["rows"], dtype=np.uint32)
        out[k + "/cols"] = np.array(c["cols"], dtype=np.uint32)
        out[k + "/values"] = np.array(c["values"], dtype=np.float64)
        out[k + "/b"] = np.array(c["b"], dtype=np.float64)
        out[k + "/params"] = np.array([c["n"], c.get("row", 0), c["seed"]], dtype=np.int64)
        out[k + "/epsilon"] = np.array([c["epsilon"]], dtype=np.float64)
        if c["kind"] == "estimate":
            out[k + "/estimates"] = np.array(r["estimates"], dtype=np.float64)
            out[k + "/mean_variance"] = np.array([r["mean"], r["variance"]], dtype=np.float64)
        else:
            out[k + "/solution"] = np.array(r["solution"], dtype=np.float64)
            out[k + "/variances"] = np.array(r["variances"], dtype=np.float64)
            out[k + "/total_variance_residual"] = np.array([r["totalVariance"], r["residual"]], dtype=np.float64)
    # the same numbers of two small cases as JSON, for the JavaScript surface test (node reads no .npz)
    js = []
    for c, r in zip(cases, res):
        if c["name"] in ("sdd40_row0_eps0.1_seed42", "sdd12_eps0.1_seed9"):
            keep = {k: c[k] for k in ("kind", "name", "n", "rows", "cols", "values", "b", "epsilon", "seed")}
            keep["row"] = c.get("row", 0)
            keep["expect"] = {k: r[k] for k in r if k not in ("name", "estimates")}
            js.append(keep)
    (ROOT / "tests" / "golden" / "reference_walk_js.json").write_text(json.dumps(js))
    path = ROOT / "tests" / "golden" / "reference_walk.npz"
    np.savez_compressed(path, **out)
    print(path, {k: v.shape for k, v in out.items() if "/estimates" in k or "/solution" in k})


if __name__ == "__main__":
    main()
