"""Oracle restatement of the CG family against the reference's own unit-test cases."""
import numpy as np

from oracle import oracle as O


def test_cg_reference_unit_case():
    # optimized_solver.rs:373-420 and fast_solver.rs:274-292: [[4,1],[1,3]] x = [1,2]
    rp, ci, va = O.csr_from_triplets([0, 0, 1, 1], [0, 1, 0, 1], [4.0, 1.0, 1.0, 3.0], 2, 2)
    r = O.cg_solve(rp, ci, va, [1.0, 2.0])
    assert r["converged"] and r["residual_norm"] < 1e-6 and r["iterations"] > 0 and r["matvec_count"] > 0
    ax = O.spmv(rp, ci, va, r["x"])
    assert np.hypot(ax[0] - 1.0, ax[1] - 2.0) < 1e-10
    np.testing.assert_allclose(r["x"], [1.0 / 11.0, 7.0 / 11.0], atol=1e-14)


def test_cg_stops_and_limits():
    rp, ci, va = O.csr_from_triplets([0, 1, 2], [0, 1, 2], [2.0, 4.0, 8.0], 3, 3)
    r = O.cg_solve(rp, ci, va, [0.0, 0.0, 0.0])
    assert r["converged"] and r["iterations"] == 0           # rsold = 0 <= tol^2 before the first matvec
    r = O.cg_solve(rp, ci, va, [2.0, 4.0, 8.0], tolerance=1e-12)
    np.testing.assert_allclose(r["x"], [1.0, 1.0, 1.0], atol=1e-12)
    assert r["iterations"] <= 3
    r = O.cg_solve(rp, ci, va, [2.0, 4.0, 8.0], tolerance=1e-300, max_iterations=1)
    assert not r["converged"] and r["iterations"] == 1


def _cg_golden():
    from pathlib import Path
    z = np.load(Path(__file__).resolve().parent / "golden" / "reference_cg.npz")
    return z, [str(c) for c in z["__cases"]], float(z["__tol"][0])


def test_g8_cg_matches_the_reference_js_twin():
    """G8: FastConjugateGradient of js/fast-solver.js:108-228 run on the reference's own SPD fixtures (tests/golden/make_golden_cg.py).
    Same algorithm, same stop rules — the JS sums its rows four entries at a time, so solutions agree to rounding, not bitwise, and
    the number of matrix-vector products may differ by the one iteration a threshold comparison lands on."""
    z, cases, tol = _cg_golden()
    for ck in cases:
        key = ck.rsplit("__", 1)[0]
        rp, ci, va, b = z[f"{key}__row_ptr"], z[f"{key}__col_idx"], z[f"{key}__values"], z[f"{ck}__b"]
        r = O.cg_solve(rp, ci, va, b, tolerance=tol, max_iterations=1000)
        xj, pj = z[f"{ck}__js_x"], int(z[f"{ck}__js_products"][0])
        scale = np.abs(xj).max()
        if pj < 100:                                   # well conditioned: the very same number of products, solutions to rounding
            assert int(r["matvec_count"]) == pj, (ck, r["matvec_count"], pj)
            assert np.abs(r["x"] - xj).max() <= 1e-9 * scale, ck
        else:                                          # ill conditioned (hundreds of iterations, or the limit): rounding moves the count a few %
            assert abs(int(r["matvec_count"]) - pj) <= pj // 20, (ck, r["matvec_count"], pj)
            assert np.abs(r["x"] - xj).max() <= 1e-5 * scale, ck
        if pj < 1000:
            assert np.linalg.norm(O.spmv(rp, ci, va, r["x"]) - b) <= 1e-8 * max(1.0, np.linalg.norm(b)), ck
