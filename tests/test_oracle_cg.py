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
