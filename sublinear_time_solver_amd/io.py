"""Matrix ingest / wire formats either side of the hot path.

File loaders follow the reference's CLI (bin/cli.js:256-270 `loadMatrix`, :449-491 parsers): `.json`
(the Matrix model of src/core/types.ts:6-22 — `dense` rows or `coo` arrays, top-level or nested under
`data`), `.csv` (dense rows), `.mtx` (MatrixMarket coordinate, 1-based, `%` comments; like the
reference's parser no symmetric expansion is applied unless asked).  Validation follows
MatrixOperations.validateMatrix (src/core/matrix.ts:11-55) and raises SolverError with the matching
variant; the triplets then go through sl_matrix_create_from_triplets, i.e. the Rust builder rules
(matrix/mod.rs:160-199: bounds, finiteness, zero dropping, stable (row, col) sort).

analyze_matrix mirrors MatrixOperations.analyzeMatrix (src/core/matrix.ts:211-351): host-side
bookkeeping (the reference computes it on the CPU as well), vectorised with scipy.
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Any, Dict, Tuple

import numpy as np

from ._lib import SolverError


# ---- files -> reference JSON matrix model -----------------------------------------------------------
def parse_csv_matrix(text: str) -> Dict[str, Any]:
    """parseCSVMatrix, bin/cli.js:449-461"""
    rows = [[float(v.strip()) for v in line.split(",")] for line in text.strip().split("\n")]
    return {"rows": len(rows), "cols": len(rows[0]), "data": rows, "format": "dense"}


def parse_matrix_market(text: str, expand_symmetric: bool = False) -> Dict[str, Any]:
    """parseMatrixMarket, bin/cli.js:463-491 (coordinate format, 1-based -> 0-based)."""
    lines = text.strip().split("\n")
    h = 0
    symmetric = False
    while lines[h].startswith("%"):
        if lines[h].lower().startswith("%%matrixmarket") and "symmetric" in lines[h].lower():
            symmetric = True
        h += 1
    rows, cols, entries = (int(float(t)) for t in lines[h].split()[:3])
    ri, ci, va = [], [], []
    for ln in lines[h + 1:]:
        if ln.strip():
            p = ln.split()
            r, c = int(p[0]) - 1, int(p[1]) - 1
            v = float(p[2]) if len(p) > 2 else 1.0
            ri.append(r), ci.append(c), va.append(v)
            if expand_symmetric and symmetric and r != c:
                ri.append(c), ci.append(r), va.append(v)
    return {"rows": rows, "cols": cols, "entries": entries, "format": "coo",
            "data": {"values": va, "rowIndices": ri, "colIndices": ci}}


def load_matrix(path: str | Path, **kw) -> Dict[str, Any]:
    """loadMatrix, bin/cli.js:256-270"""
    p = Path(path)
    if not p.exists():
        raise SolverError(4, f"Matrix file not found: {p}")
    text = p.read_text()
    ext = p.suffix.lower()
    if ext == ".json":
        return json.loads(text)
    if ext == ".csv":
        return parse_csv_matrix(text)
    if ext == ".mtx":
        return parse_matrix_market(text, **kw)
    raise SolverError(6, f"Unsupported matrix format: {ext}")


def load_vector(path: str | Path) -> np.ndarray:
    """loadVector, bin/cli.js:272-290: JSON array, or one number per line / comma separated"""
    p = Path(path)
    if not p.exists():
        raise SolverError(4, f"Vector file not found: {p}")
    text = p.read_text()
    if p.suffix.lower() == ".json":
        v = json.loads(text)
        if isinstance(v, dict):
            v = v.get("data", v.get("values"))
        if not isinstance(v, list):
            raise SolverError(4, "Vector must be an array of numbers")
        return np.asarray(v, dtype=np.float64)
    return np.asarray([float(t) for t in text.replace(",", "\n").split()], dtype=np.float64)


# ---- reference JSON matrix model -> validated triplets ------------------------------------------------
def matrix_to_triplets(matrix: Dict[str, Any]) -> Tuple[np.ndarray, np.ndarray, np.ndarray, int, int]:
    """validateMatrix (src/core/matrix.ts:11-55) then (row, col, value) arrays; dense zeros are not emitted
    (SparseMatrix::from_dense, matrix/mod.rs:204-223)."""
    if not matrix:
        raise SolverError(4, "Matrix is required")
    rows, cols = int(matrix.get("rows", 0)), int(matrix.get("cols", 0))
    if rows <= 0 or cols <= 0:
        raise SolverError(5, "Matrix dimensions must be positive")
    fmt = matrix.get("format")
    if fmt == "dense":
        data = matrix.get("data")
        if not isinstance(data, (list, np.ndarray)) or len(data) != rows:
            raise SolverError(4, "Dense matrix data must be array of rows")
        for i, row in enumerate(data):
            if len(row) != cols:
                raise SolverError(4, f"Row {i} has invalid length")
        a = np.asarray(data, dtype=np.float64)
        r, c = np.nonzero(a)
        return r.astype(np.int64), c.astype(np.int64), a[r, c], rows, cols
    if fmt == "coo":
        src = matrix["data"] if isinstance(matrix.get("data"), dict) else matrix       # bin/cli.js nests under `data`
        try:
            v, r, c = src["values"], src["rowIndices"], src["colIndices"]
        except KeyError:
            raise SolverError(4, "COO matrix must have values, rowIndices, and colIndices arrays")
        if not (len(v) == len(r) == len(c)):
            raise SolverError(4, "COO matrix arrays must have same length")
        r = np.asarray(r, dtype=np.int64)
        c = np.asarray(c, dtype=np.int64)
        v = np.asarray(v, dtype=np.float64)
        bad_r, bad_c = (r < 0) | (r >= rows), (c < 0) | (c >= cols)      # entry after entry, the row before the column (matrix.ts:44-51)
        bad = np.nonzero(bad_r | bad_c)[0]
        if bad.size:
            k = int(bad[0])
            raise SolverError(8, f"Invalid row index {int(r[k])}" if bad_r[k] else f"Invalid column index {int(c[k])}")
        return r, c, v, rows, cols
    raise SolverError(6, f"Unsupported matrix format: {fmt}")


def matrix_to_device(matrix: Dict[str, Any], **kw):
    """JSON model -> SparseMatrix in HBM (through the reference's triplet builder rules)."""
    from .solver import SparseMatrix

    r, c, v, rows, cols = matrix_to_triplets(matrix)
    return SparseMatrix.from_triplets(zip(r.tolist(), c.tolist(), v.tolist()), rows, cols, **kw)


# ---- analyzeMatrix -------------------------------------------------------------------------------
def _ordered_group_sums(group: np.ndarray, values: np.ndarray, n: int) -> np.ndarray:
    """out[g] = the values of group g added one after the other IN THE ORDER GIVEN (a running sum per group, as a `sum +=` loop over the
    stored entries forms it) — vectorised over the groups, sequential inside each"""
    out = np.zeros(n)
    if not values.size:
        return out
    order = np.argsort(group, kind="stable")
    g, val = group[order], values[order]
    start = np.zeros(n + 1, dtype=np.int64)
    np.add.at(start, g + 1, 1)
    start = np.cumsum(start)
    lens = np.diff(start)
    for k in range(int(lens.max())):
        rows = np.flatnonzero(lens > k)
        out[rows] = out[rows] + val[start[rows] + k]
    return out


def analyze_matrix(matrix: Dict[str, Any]) -> Dict[str, Any]:
    """MatrixOperations.analyzeMatrix (src/core/matrix.ts:327-351) with checkDiagonalDominance (:211-258), isSymmetric (:263-296,
    tolerance 1e-10) and calculateSparsity (:301-322) — with the reference's own reading of the matrix, so that every field (the bits of
    dominanceStrength included) is what its TypeScript returns (golden G14, tests/golden/make_golden_ts_analyze.py):
    the DIAGONAL and the symmetry test read entries through getEntry, i.e. the FIRST stored match of a duplicated COO entry (:105-112); the
    off-diagonal row / column sums add |value| over ALL stored entries in storage order (getRowSum / getColumnSum, :143-206); the loop over
    the rows stops at the first zero diagonal with everything false and strength 0 (:228-234)."""
    import scipy.sparse as sp

    r, c, v, rows, cols = matrix_to_triplets(matrix)
    is_row, is_col, strength, symmetric = False, False, 0.0, False
    if rows == cols:
        n = rows
        _, first = np.unique(r * n + c, return_index=True)               # the first stored match of every (row, col)
        fr, fc, fv = r[first], c[first], v[first]
        d = np.zeros(n)
        on = fr == fc
        d[fr[on]] = np.abs(fv[on])
        off = r != c
        row_off = _ordered_group_sums(r[off], np.abs(v[off]), n)
        col_off = _ordered_group_sums(c[off], np.abs(v[off]), n)
        if not (d == 0).any():
            rs, cs = d - row_off, d - col_off
            is_row, is_col = bool((rs >= 0).all()), bool((cs >= 0).all())
            min_r = float((rs[rs >= 0] / d[rs >= 0]).min()) if (rs >= 0).any() else float("inf")
            min_c = float((cs[cs >= 0] / d[cs >= 0]).min()) if (cs >= 0).any() else float("inf")
            strength = max(min_r if is_row else 0.0, min_c if is_col else 0.0)
        F = sp.csr_matrix((fv, (fr, fc)), shape=(n, n))                  # the matrix getEntry sees
        symmetric = (abs(F - F.T) > 1e-10).nnz == 0
    if matrix.get("format") == "dense":
        sparsity = 1.0 - float((np.abs(v) > 1e-15).sum()) / (rows * cols)
    else:
        sparsity = 1.0 - len(v) / (rows * cols)
    dtype = "row" if is_row else ("column" if is_col else "none")
    return {"isDiagonallyDominant": is_row or is_col, "dominanceType": dtype, "dominanceStrength": strength,
            "isSymmetric": bool(symmetric), "sparsity": sparsity, "size": {"rows": rows, "cols": cols}}


def generate_matrix(kind: str, size: int, seed: int = 42, **kw) -> Dict[str, Any]:
    """`generate -t <type> -s <size>` (src/cli/index.ts:308-352, src/mcp/tools/matrix.ts:297-322), seeded."""
    from . import generators as G

    if kind == "diagonally-dominant":
        rp, ci, va, _ = G.gen1000_dense(size=size, strength=kw.get("strength", 2.0), seed=seed)
        a = np.zeros((size, size))
        for i in range(size):
            a[i, ci[rp[i]:rp[i + 1]]] = va[rp[i]:rp[i + 1]]
        return {"rows": size, "cols": size, "format": "dense", "data": a.tolist()}
    if kind == "banded":
        rp, ci, va, _ = G.sdd_rows(size, kw.get("k", 8), seed, kw.get("bandwidth", max(8, size // 10)))
    elif kind == "sparse":
        rp, ci, va, _ = G.sdd_rows(size, kw.get("k", 8), seed, 0)
    else:
        raise SolverError(4, f"Unknown matrix type: {kind}")
    rows = np.repeat(np.arange(size), np.diff(rp.astype(np.int64)))
    return {"rows": size, "cols": size, "format": "coo", "values": va.tolist(), "rowIndices": rows.tolist(), "colIndices": ci.tolist()}
