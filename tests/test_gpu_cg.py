"""Conjugate-gradient path behind the same SpMV: GPU vs the oracle restatement of
OptimizedConjugateGradientSolver::solve (src/optimized_solver.rs:182-295).  Dots are tree reductions on the
device, so the tolerance is 1e-10 relative (stated), not bitwise."""
import numpy as np
import pytest

import sublinear_time_solver_amd as S
from sublinear_time_solver_amd import _lib as L
from sublinear_time_solver_amd import generators as G
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _spd_banded(n, k=7, seed=0):
    """symmetric strictly dominant => SPD: A = D + (B + B^T)"""
    rng = np.random.default_rng(seed)
    import scipy.sparse as sp
    B = sp.random(n, n, density=k / n, random_state=rng, data_rvs=lambda s: rng.uniform(-1, 1, s), format="csr")
    A = B + B.T
    d = np.abs(A).sum(axis=1).A1 + 1.0
    A = (A + sp.diags(d)).tocsr()
    A.sort_indices()
    return A.indptr.astype(np.uint32), A.indices.astype(np.uint32), A.data.astype(np.float64)


def test_cg_kat_from_reference_test(gpu):
    # optimized_solver.rs:401-420 (test_optimized_conjugate_gradient): [[4,1],[1,3]] x = [1,2]
    m = S.SparseMatrix.from_triplets([(0, 0, 4.0), (0, 1, 1.0), (1, 0, 1.0), (1, 1, 3.0)], 2, 2)
    r = S.ConjugateGradientSolver().solve(m, [1.0, 2.0])
    assert r.converged and r.residual_norm < 1e-6 and r.iterations > 0
    ax = m.multiply_vector(r.solution)
    assert np.hypot(ax[0] - 1.0, ax[1] - 2.0) < 1e-10
    np.testing.assert_allclose(r.solution, [1.0 / 11.0, 7.0 / 11.0], atol=1e-12)
    with pytest.raises(S.SolverError) as e:
        S.ConjugateGradientSolver().solve(m, [1.0])
    assert e.value.kind == "DimensionMismatch"


@pytest.mark.parametrize("n,order", [(5000, 0), (5000, 1), (60000, 0)])
def test_cg_matches_oracle(gpu, n, order):
    rp, ci, va = _spd_banded(n, seed=n)
    b = 1.0 + np.cos(np.arange(n))
    m = S.SparseMatrix.from_csr(rp, ci, va, n, n)
    # tolerance^2 must stay well above the reference's |p.Ap| < 1e-16 bail-out (optimized_solver.rs:236-238),
    # otherwise the loop exits unconverged on either side depending on the last bit of a dot product
    g = S.ConjugateGradientSolver(tolerance=1e-7, order=order).solve(m, b)
    o = O.cg_solve(rp, ci, va, b, tolerance=1e-7, order=order)
    assert g.converged and o["converged"]
    assert abs(g.iterations - o["iterations"]) <= 1          # a tree-reduced dot may cross the stop rule one step apart
    if g.iterations == o["iterations"]:
        assert np.max(np.abs(g.solution - o["x"])) <= 1e-10 * np.max(np.abs(o["x"]))
    assert np.linalg.norm(O.spmv(rp, ci, va, g.solution) - b) <= 2e-7
    g2 = S.ConjugateGradientSolver(tolerance=1e-7, order=order).solve(m, b)
    assert (g.solution == g2.solution).all()                 # run-to-run determinism


def test_cg_on_sdd_system_agrees_with_neumann(gpu):
    """the headline system is row dominant but NOT symmetric: CG is not guaranteed there; on a symmetric
    dominant system both methods must agree"""
    n = 20000
    rp, ci, va = _spd_banded(n, seed=1)
    b = np.ones(n)
    m = S.SparseMatrix.from_csr(rp, ci, va, n, n)
    c = S.ConjugateGradientSolver(tolerance=1e-7).solve(m, b)
    assert c.converged
    nm = S.NeumannSolver(max_terms=500, series_tolerance=1e-14).solve(m, b, S.SolverOptions(tolerance=1e-11))
    np.testing.assert_allclose(c.solution, nm.solution, atol=1e-7)


def test_g8_gpu_cg_matches_the_reference_js_twin(gpu):
    """the GPU CG against the golden vectors of the reference's runnable JS FastConjugateGradient (tests/golden/reference_cg.npz)"""
    from pathlib import Path
    z = np.load(Path(__file__).resolve().parent / "golden" / "reference_cg.npz")
    tol = float(z["__tol"][0])
    for ck in (str(c) for c in z["__cases"]):
        key = ck.rsplit("__", 1)[0]
        rp, ci, va, b = z[f"{key}__row_ptr"], z[f"{key}__col_idx"], z[f"{key}__values"], z[f"{ck}__b"]
        n = b.size
        m = S.SparseMatrix.from_csr(rp, ci, va, n, n)
        r = S.ConjugateGradientSolver(max_iterations=1000, tolerance=tol).solve(m, b)
        xj, pj = z[f"{ck}__js_x"], int(z[f"{ck}__js_products"][0])
        scale = np.abs(xj).max()
        if pj < 100:
            assert abs(int(r.stats["matvec_count"]) - pj) <= 1, (ck, r.stats["matvec_count"], pj)     # tree-reduced dots: +-1 at a threshold
            assert np.abs(r.solution - xj).max() <= 1e-9 * scale, ck
        else:
            assert abs(int(r.stats["matvec_count"]) - pj) <= pj // 20, (ck, r.stats["matvec_count"], pj)
            assert np.abs(r.solution - xj).max() <= 1e-5 * scale, ck
