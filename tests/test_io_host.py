"""Wire formats either side of the path: loaders and validation follow bin/cli.js:256-491 and
src/core/matrix.ts:11-55,211-351; the triplets they produce go through the reference's builder rules."""
import json

import numpy as np
import pytest

from sublinear_time_solver_amd import io, SolverError
from oracle import oracle as O

DENSE = {"rows": 3, "cols": 3, "format": "dense", "data": [[4, -1, 0], [-1, 4, -1], [0, -1, 3]]}
COO = {"rows": 3, "cols": 3, "format": "coo", "values": [4, -1, -1, 4, -1, -1, 3],
       "rowIndices": [0, 0, 1, 1, 1, 2, 2], "colIndices": [0, 1, 0, 1, 2, 1, 2]}


def _csr(matrix):
    r, c, v, rows, cols = io.matrix_to_triplets(matrix)
    return O.csr_from_triplets(r, c, v, rows, cols)


def test_all_layouts_yield_the_same_csr(tmp_path):
    ref = _csr(DENSE)
    nested = {"rows": 3, "cols": 3, "format": "coo", "data": {k: COO[k] for k in ("values", "rowIndices", "colIndices")}}
    (tmp_path / "a.json").write_text(json.dumps(COO))
    (tmp_path / "a.csv").write_text("4,-1,0\n-1,4,-1\n0,-1,3\n")
    (tmp_path / "a.mtx").write_text("%%MatrixMarket matrix coordinate real general\n% comment\n3 3 7\n1 1 4\n1 2 -1\n2 1 -1\n2 2 4\n2 3 -1\n3 2 -1\n3 3 3\n")
    for m in (COO, nested, io.load_matrix(tmp_path / "a.json"), io.load_matrix(tmp_path / "a.csv"), io.load_matrix(tmp_path / "a.mtx")):
        got = _csr(m)
        assert got[0].tolist() == ref[0].tolist() and got[1].tolist() == ref[1].tolist() and got[2].tolist() == ref[2].tolist()
    mm = io.load_matrix(tmp_path / "a.mtx")
    assert mm["entries"] == 7 and mm["format"] == "coo" and mm["data"]["rowIndices"][0] == 0        # 1-based -> 0-based


def test_matrix_market_symmetric_expansion_is_opt_in(tmp_path):
    (tmp_path / "s.mtx").write_text("%%MatrixMarket matrix coordinate real symmetric\n2 2 2\n1 1 2.0\n2 1 0.5\n")
    plain = io.load_matrix(tmp_path / "s.mtx")                       # like the reference's parser: entries as stored
    assert len(plain["data"]["values"]) == 2
    full = io.load_matrix(tmp_path / "s.mtx", expand_symmetric=True)
    assert sorted(zip(full["data"]["rowIndices"], full["data"]["colIndices"])) == [(0, 0), (0, 1), (1, 0)]


def test_validation_errors_match_the_reference_rules(tmp_path):
    with pytest.raises(SolverError) as e:
        io.matrix_to_triplets({"rows": 0, "cols": 3, "format": "dense", "data": []})
    assert e.value.kind == "DimensionMismatch"                       # 'Matrix dimensions must be positive'
    with pytest.raises(SolverError) as e:
        io.matrix_to_triplets({"rows": 2, "cols": 2, "format": "dense", "data": [[1, 2], [3]]})
    assert "Row 1 has invalid length" in str(e.value)
    with pytest.raises(SolverError) as e:
        io.matrix_to_triplets({"rows": 2, "cols": 2, "format": "coo", "values": [1], "rowIndices": [0, 1], "colIndices": [0]})
    assert "same length" in str(e.value)
    with pytest.raises(SolverError) as e:
        io.matrix_to_triplets({"rows": 2, "cols": 2, "format": "coo", "values": [1.0], "rowIndices": [2], "colIndices": [0]})
    assert e.value.kind == "IndexOutOfBounds" and "Invalid row index 2" in str(e.value)
    with pytest.raises(SolverError) as e:
        io.matrix_to_triplets({"rows": 2, "cols": 2, "format": "csc"})
    assert e.value.kind == "UnsupportedMatrixFormat"
    with pytest.raises(SolverError) as e:
        io.load_matrix(tmp_path / "nope.json")
    assert "not found" in str(e.value)
    (tmp_path / "a.xyz").write_text("1")
    with pytest.raises(SolverError) as e:
        io.load_matrix(tmp_path / "a.xyz")
    assert e.value.kind == "UnsupportedMatrixFormat"


def test_analyze_matrix_fields():
    a = io.analyze_matrix(DENSE)
    assert a["isDiagonallyDominant"] and a["dominanceType"] == "row" and a["isSymmetric"]
    assert abs(a["dominanceStrength"] - 0.5) < 1e-12 and abs(a["sparsity"] - 2 / 9) < 1e-12      # min (d - off)/d = (4-2)/4
    assert a["size"] == {"rows": 3, "cols": 3}
    col_dd = {"rows": 2, "cols": 2, "format": "dense", "data": [[1.0, -0.9], [-0.1, 1.0]]}           # PageRank-like: column dominant
    col_dd["data"][0][1] = -2.0
    b = io.analyze_matrix(col_dd)
    assert b["isDiagonallyDominant"] is False or b["dominanceType"] in ("column", "none")
    pr = {"rows": 2, "cols": 2, "format": "dense", "data": [[1.0, -0.8], [-0.5, 1.0]]}
    assert io.analyze_matrix(pr)["dominanceType"] == "row"
    zero_diag = {"rows": 2, "cols": 2, "format": "dense", "data": [[0.0, 1.0], [1.0, 2.0]]}
    z = io.analyze_matrix(zero_diag)
    assert not z["isDiagonallyDominant"] and z["dominanceStrength"] == 0


def test_generate_and_vectors(tmp_path):
    g = io.generate_matrix("diagonally-dominant", 30, seed=42)
    assert g["format"] == "dense" and len(g["data"]) == 30 and io.analyze_matrix(g)["dominanceType"] == "row"
    g2 = io.generate_matrix("diagonally-dominant", 30, seed=42)
    assert g == g2
    s = io.generate_matrix("sparse", 200, seed=1)
    assert s["format"] == "coo" and len(s["values"]) == 200 * 8
    (tmp_path / "b.json").write_text("[1, 2, 3]")
    (tmp_path / "b.txt").write_text("1\n2\n3\n")
    assert io.load_vector(tmp_path / "b.json").tolist() == io.load_vector(tmp_path / "b.txt").tolist() == [1.0, 2.0, 3.0]


def test_g14_analyze_matrix_equals_the_references_own_typescript():
    """MatrixOperations.analyzeMatrix (core/matrix.ts:327-351) as the reference's own code evaluated it on 13 matrices
    (tests/golden/make_golden_ts_analyze.py): every field equal — the bits of dominanceStrength, the diagonal read by FIRST stored match,
    row / column sums over all duplicates in storage order, the stop at the first zero diagonal, symmetry through getEntry"""
    import json
    from pathlib import Path
    from sublinear_time_solver_amd import io
    cases = json.loads((Path(__file__).resolve().parent / "golden" / "reference_ts_analyze.json").read_text())
    assert len(cases) >= 14
    from sublinear_time_solver_amd import SolverError
    valid = [c for c in cases if not c.get("invalid")]
    for c in valid:
        assert io.analyze_matrix(c["matrix"]) == c["analysis"], c["name"]
    assert {c["analysis"]["dominanceType"] for c in valid} >= {"row", "column", "none"} and any(c["analysis"]["isSymmetric"] for c in valid)
    # validateMatrix (core/matrix.ts:11-55): the same message for the same malformed matrix — entry after entry, the row before the column
    bad = [c for c in cases if c.get("invalid")]
    assert len(bad) >= 8
    for c in bad:
        with pytest.raises(SolverError) as e:
            io.analyze_matrix(c["matrix"])
        assert c["error"]["message"] in str(e.value), (c["name"], str(e.value))
