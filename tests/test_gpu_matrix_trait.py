"""The rest of `trait Matrix` and of NeumannState that round 4's verdict found missing, through the C ABI against the oracle:
  Matrix::multiply_vector_add        matrix/mod.rs:441-465 over CSRStorage::multiply_vector_add, sparse.rs:192-203   (bit-exact)
  Matrix::diagonal_dominance_factor  matrix/mod.rs:487-514                                                           (bit-exact)
  Matrix::spectral_radius_estimate   matrix/mod.rs:83-100                                                            (bit-exact)
  NeumannState::estimate_error_bounds neumann.rs:321-347   (<= 1e-12 relative: the device's norms are tree-reduced; None / Some(0.0) cases exact)
Nothing here reads /root/reference."""
import numpy as np
import pytest

import sublinear_time_solver_amd as S
from sublinear_time_solver_amd import _lib as L
from sublinear_time_solver_amd import generators as G
from oracle import oracle as O
from tests.test_gpu_longrows import _hub_system

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def _ragged(n=1500, seed=3):
    """rows of 0..40 entries (some empty, some without a diagonal entry), columns anywhere, values of mixed magnitude"""
    rng = np.random.default_rng(seed)
    tr, tc, tv = [], [], []
    for i in range(n):
        m = int(rng.integers(0, 41)) if i % 11 else 0
        cols = np.sort(rng.choice(n, size=m, replace=False))
        vals = rng.standard_normal(m) * 10.0 ** rng.integers(-3, 4, size=m)
        tr += [i] * m; tc += cols.tolist(); tv += vals.tolist()
    return O.csr_from_triplets(tr, tc, tv, n, n)


SYSTEMS = {
    "sdd16": lambda: G.sdd_rows(5000 + 13, 16, seed=4)[:3],
    "sdd8_band": lambda: G.sdd_rows(4096, 8, seed=2, half_bandwidth=64)[:3],
    "ragged": _ragged,
    "hubs": lambda: _hub_system(n=3000, seed=7),
}


@pytest.mark.parametrize("name", sorted(SYSTEMS))
def test_multiply_vector_add_keeps_the_reference_rounding(gpu, name):
    rp, ci, va = SYSTEMS[name]()
    n = rp.size - 1
    m = S.SparseMatrix.from_csr(rp, ci, va, n, n)
    rng = np.random.default_rng(11)
    x = rng.standard_normal(n)
    y0 = rng.standard_normal(n) * 3.0
    want = O.spmv_add(rp, ci, va, x, y0)
    got = y0.copy()
    assert m.multiply_vector_add(x, got) is got
    bad = np.nonzero(bits(got) != bits(want))[0]
    assert bad.size == 0, f"{bad.size} of {n} rows differ, first {bad[:5]}"
    # what makes it a function of its own: product-then-add rounds differently on most long enough rows
    two_step = y0 + O.spmv(rp, ci, va, x)
    if name != "ragged":
        assert (bits(two_step) != bits(want)).any()
    # y = 0: the accumulating form IS multiply_vector (sparse.rs:187-190 is written that way)
    z = np.zeros(n)
    m.multiply_vector_add(x, z)
    assert (bits(z) == bits(m.multiply_vector(x))).all()


def test_multiply_vector_add_on_a_row_slice_and_its_errors(gpu):
    rp, ci, va, _ = G.sdd_rows(3000, 16, seed=9)
    n, lo, hi = 3000, 1000, 2200
    srp = (rp[lo:hi + 1] - rp[lo]).astype(np.uint32)
    sci, sva = ci[rp[lo]:rp[hi]], va[rp[lo]:rp[hi]]
    m = S.SparseMatrix.from_csr(srp, sci, sva, hi - lo, n, row_offset=lo)
    x = np.cos(np.arange(n) * 0.01)
    y = np.linspace(-1.0, 1.0, hi - lo)
    want = O.spmv_add(srp, sci, sva, x, y)
    m.multiply_vector_add(x, y)
    assert (bits(y) == bits(want)).all()
    with pytest.raises(S.SolverError) as e:
        m.multiply_vector_add(x[:-1], y)
    assert e.value.kind == "DimensionMismatch"
    with pytest.raises(S.SolverError) as e:
        m.multiply_vector_add(x, np.zeros(5))
    assert e.value.kind == "DimensionMismatch"
    lib = L.load()
    assert lib.sl_spmv_add(m._h, L.ptr(x), L.ptr(y), L.SL_ORDER_SIMD4, L.SL_MEM_HOST) == 4       # no accumulating simd form in the reference


@pytest.mark.parametrize("name", sorted(SYSTEMS))
def test_conditioning_info_bit_for_bit(gpu, name):
    rp, ci, va = SYSTEMS[name]()
    n = rp.size - 1
    m = S.SparseMatrix.from_csr(rp, ci, va, n, n)
    f, of = m.diagonal_dominance_factor(), O.diagonal_dominance_factor(rp, ci, va)
    assert (f is None) == (of is None)
    if f is not None:
        assert bits([f])[0] == bits([of])[0], (f, of)
    assert bits([m.spectral_radius_estimate()])[0] == bits([O.spectral_radius_estimate(rp, ci, va)])[0]
    info = m.conditioning_info()
    assert info["is_diagonally_dominant"] == O.is_diagonally_dominant(rp, ci, va) and info["condition_number"] is None


def test_dominance_factor_known_answers(gpu):
    m = S.SparseMatrix.from_triplets([(0, 0, 4.0), (0, 1, 1.0), (1, 0, 1.0), (1, 1, 3.0)], 2, 2)
    assert m.diagonal_dominance_factor() == 3.0 and m.spectral_radius_estimate() == 5.0
    d = S.SparseMatrix.from_triplets([(0, 0, 2.0), (1, 1, 3.0)], 2, 2)                     # no off-diagonal weight anywhere: None (mod.rs:508-512)
    assert d.diagonal_dominance_factor() is None and d.spectral_radius_estimate() == 3.0
    z = S.SparseMatrix.from_triplets([(0, 1, 2.0), (1, 1, 3.0)], 2, 2)                     # a row without a diagonal entry: factor 0
    assert z.diagonal_dominance_factor() == 0.0
    e = S.SparseMatrix.from_triplets([], 3, 3)
    assert e.diagonal_dominance_factor() is None and e.spectral_radius_estimate() == 0.0


def test_error_bound_of_the_neumann_truncation(gpu):
    # G4 system: the series converges in 17 terms, rho(M) = sqrt(1/12); oracle KAT 1.724974623182487e-09
    m = S.SparseMatrix.from_triplets([(0, 0, 4.0), (0, 1, 1.0), (1, 0, 1.0), (1, 1, 3.0)], 2, 2)
    opt = S.SolverOptions(tolerance=1e-30, compute_error_bounds=True)
    r = S.NeumannSolver(20, 1e-8).solve(m, [5.0, 4.0], opt)
    o = O.neumann_solve([0, 2, 4], [0, 1, 0, 1], [4.0, 1, 1, 3], [5.0, 4.0], tolerance=1e-30, max_terms=20)
    assert r.iterations == o["iterations"] == 17 and o["error_bound"] is not None
    assert r.error_bounds is not None and abs(r.error_bounds - o["error_bound"]) <= 1e-12 * o["error_bound"]
    assert abs(o["error_bound"] - 1.724974623182487e-09) < 1e-22
    # one term computed: matrix_norm_estimate stays 0.0 => Some(0.0) (neumann.rs:327-344)
    one = S.NeumannSolver(20, 1e9).solve(m, [5.0, 4.0], opt)
    assert one.iterations == 1 and one.error_bounds == 0.0
    # residual test ends the solve before the series does: the reference leaves None; flag off: None
    assert S.NeumannSolver(20, 1e-8).solve(m, [5.0, 4.0], S.SolverOptions(tolerance=1e-6, compute_error_bounds=True)).error_bounds is None
    assert S.NeumannSolver(20, 1e-8).solve(m, [5.0, 4.0], S.SolverOptions(tolerance=1e-30)).error_bounds is None
    # adaptive_truncation off: the reference never calls estimate_error_bounds (neumann.rs:494-496)
    assert S.NeumannSolver(20, 1e-8, adaptive_truncation=False).solve(m, [5.0, 4.0], opt).error_bounds is None


@pytest.mark.parametrize("n,k,seed", [(20000, 16, 1), (4099, 8, 5)])
def test_error_bound_on_sdd_systems(gpu, n, k, seed):
    rp, ci, va, b = G.sdd_rows(n, k, seed=seed)
    m = S.SparseMatrix.from_csr(rp, ci, va, n, n)
    r = S.NeumannSolver().solve(m, b, S.SolverOptions(tolerance=1e-30, compute_error_bounds=True))
    o = O.neumann_solve(rp, ci, va, b, tolerance=1e-30)
    assert r.iterations == o["iterations"] and o["series_converged"] and o["error_bound"] is not None
    assert abs(r.error_bounds - o["error_bound"]) <= 1e-12 * o["error_bound"]
    # and it IS a bound on what the truncation left out: ||x - x_inf||_2 for the exact series limit, here taken from a much longer series
    long = O.neumann_solve(rp, ci, va, b, tolerance=1e-30, series_tolerance=1e-300, max_terms=200, max_iterations=200, raise_on_error=False)
    assert np.linalg.norm(o["x"] - long["x"]) <= o["error_bound"]
