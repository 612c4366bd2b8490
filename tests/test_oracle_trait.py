"""Pins of the oracle twins added in round 5 (CPU only).  The reference holds no unit test for these four functions
(grep of /root/reference: multiply_vector_add, diagonal_dominance_factor, spectral_radius_estimate and estimate_error_bounds are
called, never tested), so they are pinned by hand-derived known answers and by independent arithmetic:
  orc_spmv_add_csr_sequential   sparse.rs:192-203       exact: the seeded chain against Python floats evaluated in the same order
  orc_diagonal_dominance_factor matrix/mod.rs:487-514   exact: against a Python loop over the reference's committed fixtures
  orc_spectral_radius_estimate  matrix/mod.rs:83-100
  orc_powi                      compiler-rt __powidf2 (what f64::powi lowers to): against exact rational arithmetic
  orc_neumann_error_bound       neumann.rs:321-347      analytic: 2x2 system with rho(M) = sqrt(1/12)"""
from fractions import Fraction
from pathlib import Path

import numpy as np

from oracle import oracle as O

GOLD = np.load(Path(__file__).parent / "golden" / "reference_jacobi.npz")


def test_spmv_add_is_the_seeded_chain_not_product_then_add():
    # one row [1, 1] . x = [1, 1] accumulated into 1e16: (1e16 + 1) + 1 = 1e16 (each add rounds back), while 1e16 + (1 + 1) = 1e16 + 2
    rp, ci, va = [0, 2], [0, 1], [1.0, 1.0]
    assert O.spmv_add(rp, ci, va, [1.0, 1.0], [1e16])[0] == 1e16
    assert 1e16 + O.spmv(rp, ci, va, [1.0, 1.0])[0] == 1e16 + 2.0 != 1e16
    # a random ragged matrix against Python floats in the reference's order
    rng = np.random.default_rng(5)
    n = 200
    cnt = rng.integers(0, 9, size=n)
    rp = np.zeros(n + 1, dtype=np.uint32); rp[1:] = np.cumsum(cnt)
    ci = rng.integers(0, n, size=rp[-1]).astype(np.uint32)
    va = rng.standard_normal(rp[-1]) * 10.0 ** rng.integers(-4, 5, size=rp[-1])
    x, y = rng.standard_normal(n), rng.standard_normal(n) * 100.0
    want = []
    for i in range(n):
        s = float(y[i])
        for k in range(rp[i], rp[i + 1]):
            s = s + float(va[k]) * float(x[ci[k]])
        want.append(s)
    got = O.spmv_add(rp, ci, va, x, y)
    assert (got.view(np.uint64) == np.array(want).view(np.uint64)).all()
    # from zeros it is multiply_vector (sparse.rs:187-190)
    assert (O.spmv_add(rp, ci, va, x, np.zeros(n)).view(np.uint64) == O.spmv(rp, ci, va, x).view(np.uint64)).all()


def _row_stats(rp, ci, va):
    out = []
    for i in range(len(rp) - 1):
        d, off = 0.0, 0.0
        for k in range(rp[i], rp[i + 1]):
            if ci[k] == i:
                d = abs(float(va[k]))
            else:
                off = off + abs(float(va[k]))
        out.append((d, off))
    return out


def test_conditioning_twins_on_the_reference_fixtures_and_known_answers():
    keys = sorted({str(c).rsplit("__", 1)[0] for c in GOLD["__cases"]}) + ["neg_n_100_sparse_dd"]
    for key in keys:
        rp, ci, va = GOLD[f"{key}__row_ptr"], GOLD[f"{key}__col_idx"], GOLD[f"{key}__values"]
        st = _row_stats(rp, ci, va)
        ratios = [d / off for d, off in st if off > 0.0]
        assert O.diagonal_dominance_factor(rp, ci, va) == min(ratios)
        assert O.spectral_radius_estimate(rp, ci, va) == max(d + off for d, off in st)
        assert (min(ratios) >= 1.0) == O.is_diagonally_dominant(rp, ci, va)            # the two reference methods agree on what dominance is
    assert O.diagonal_dominance_factor([0, 2, 4], [0, 1, 0, 1], [4.0, 1.0, 1.0, 3.0]) == 3.0
    assert O.diagonal_dominance_factor([0, 1, 2], [0, 1], [2.0, 3.0]) is None          # no off-diagonal weight: None (mod.rs:508-512)
    assert O.diagonal_dominance_factor([0, 1, 2], [1, 1], [2.0, 3.0]) == 0.0           # a row without a diagonal entry
    assert O.diagonal_dominance_factor([0, 0, 0], [], []) is None
    assert O.spectral_radius_estimate([0, 0, 0], [], []) == 0.0


def test_powi_is_compiler_rt_square_and_multiply():
    assert O.powi(0.5, 7) == 0.0078125 and O.powi(2.0, -3) == 0.125 and O.powi(3.0, 0) == 1.0 and O.powi(0.0, 1) == 0.0
    # the sequence of roundings: r and a are updated as in __powidf2; mirror it with Python floats
    for a0 in (0.2886751345948129, 0.9999999, 1.0000001, 0.1):
        for b in (1, 2, 3, 5, 17, 34, 50):
            r, a, e = 1.0, a0, b
            while True:
                if e & 1:
                    r = r * a
                e //= 2
                if e == 0:
                    break
                a = a * a
            assert O.powi(a0, b) == r
            exact = Fraction(a0) ** b
            assert abs(Fraction(O.powi(a0, b)) - exact) <= exact * Fraction(b, 2 ** 52)   # and it is the power, to a few ulps


def test_error_bound_analytic_case_and_the_none_and_zero_cases():
    rp, ci, va, b = [0, 2, 4], [0, 1, 0, 1], [4.0, 1.0, 1.0, 3.0], [5.0, 4.0]
    r = O.neumann_solve(rp, ci, va, b, tolerance=1e-30, max_terms=20)
    assert r["iterations"] == 17 and r["series_converged"]
    # M = I - D^-1 A has eigenvalues +-sqrt(1/12): terms shrink by exactly 1/12 every two steps, so the norm estimate is rho to rounding
    rho = (1.0 / 12.0) ** 0.5
    rhs = np.array([5.0 / 4.0, 4.0 / 3.0])
    want = rho ** 17 / (1.0 - rho) * np.linalg.norm(rhs)
    assert abs(r["error_bound"] - want) <= 0.05 * want          # the estimate (||t_16|| / ||rhs||)^(1/16) sees the two eigenvector components
    assert abs(r["error_bound"] - 1.724974623182487e-09) < 1e-22
    assert np.linalg.norm(r["x"] - np.array([1.0, 1.0])) <= r["error_bound"]         # and it bounds the truncation error
    one = O.neumann_solve(rp, ci, va, b, tolerance=1e-30, series_tolerance=1e9)
    assert one["terms"] == 1 and one["error_bound"] == 0.0                            # matrix_norm_estimate stays 0.0 => Some(0.0)
    assert O.neumann_solve(rp, ci, va, b, tolerance=1e-6)["error_bound"] is None      # residual test first: series_converged false => None
    assert O.neumann_error_bound([0.0, 0.0], [0.0, 0.0], 3, True) is None             # 0 / 0 = NaN estimate: `NaN < 1.0` is false => None
    assert O.neumann_error_bound([3.0, 4.0], [3.0, 4.0], 2, True) is None             # estimate 1.0: not < 1.0 => None
    assert O.neumann_error_bound([1.0], [2.0], 0, True) is None and O.neumann_error_bound([1.0], [2.0], 5, False) is None


# ---- the element / iterator / norm side of trait Matrix (ABI version 4): the oracle twins the GPU tests compare against ----

def test_get_row_col_known_answers_of_the_reference():
    """sparse.rs:910-920 (test_csr_creation) + the bounds rule of SparseMatrix::get (matrix/mod.rs:384-386)"""
    rp, ci, va = O.csr_from_triplets([0, 0, 1, 2, 2], [0, 2, 1, 0, 2], [1.0, 2.0, 3.0, 4.0, 5.0], 3, 3)
    assert O.matrix_get(rp, ci, va, 0, 0) == 1.0 and O.matrix_get(rp, ci, va, 0, 2) == 2.0 and O.matrix_get(rp, ci, va, 1, 1) == 3.0
    assert O.matrix_get(rp, ci, va, 0, 1) is None and O.matrix_get(rp, ci, va, 3, 0) is None and O.matrix_get(rp, ci, va, 0, 3) is None
    co, vo = O.csr_row(rp, ci, va, 2)
    assert co.tolist() == [0, 2] and vo.tolist() == [4.0, 5.0] and O.csr_row(rp, ci, va, 3)[0].size == 0
    ro, wo = O.csr_col(rp, ci, va, 0)
    assert ro.tolist() == [0, 2] and wo.tolist() == [1.0, 4.0] and O.csr_col(rp, ci, va, 1)[0].tolist() == [1]
    assert O.frobenius_norm(rp, va) == np.sqrt(55.0)
    assert O.sparsity_info(rp, ci) == {"nnz": 5, "dimensions": (3, 3), "sparsity_ratio": 5.0 / 9.0, "avg_nnz_per_row": 5.0 / 3.0,
                                       "max_nnz_per_row": 2, "bandwidth": 2, "is_banded": False}


def test_a_duplicated_column_answers_with_the_entry_the_halving_search_lands_on():
    """from_triplets keeps duplicates as entries of their own (sparse.rs:80-132, stable sort); get is
    `col_indices[start..end].binary_search(&col)` (sparse.rs:150): mid = lo + (hi - lo) / 2 until the column is met"""
    # row 0: columns [1, 3, 3, 3, 3, 7] with values 10..15: lo = 0, hi = 6 -> mid 3 (column 3) -> the THIRD of the four, value 13
    rp, ci, va = O.csr_from_triplets([0] * 6, [1, 3, 3, 3, 3, 7], [10.0, 11.0, 12.0, 13.0, 14.0, 15.0], 1, 8)
    assert ci.tolist() == [1, 3, 3, 3, 3, 7] and va.tolist() == [10.0, 11.0, 12.0, 13.0, 14.0, 15.0]
    assert O.matrix_get(rp, ci, va, 0, 3, cols=8) == 13.0
    # col_iter searches each row once: ONE pair for the row, the same entry (CSRColIter::next, sparse.rs:282-296)
    ro, wo = O.csr_col(rp, ci, va, 3)
    assert ro.tolist() == [0] and wo.tolist() == [13.0]
    # row_iter yields all of them, in input order
    assert O.csr_row(rp, ci, va, 0)[1].tolist() == va.tolist()
    # the Frobenius norm counts every stored entry, duplicates included
    assert O.frobenius_norm(rp, va) == np.sqrt(sum(v * v for v in va))


def test_sparsity_info_edge_cases():
    rp, ci, _ = O.csr_from_triplets([], [], [], 4, 6)
    assert O.sparsity_info(rp, ci, cols=6) == {"nnz": 0, "dimensions": (4, 6), "sparsity_ratio": 0.0, "avg_nnz_per_row": 0.0, "max_nnz_per_row": 0,
                                               "bandwidth": 0, "is_banded": True}          # 0 < 4 / 4 (matrix/mod.rs:542, integer division)
    rp, ci, _ = O.csr_from_triplets([0, 7], [0, 6], [1.0, 1.0], 8, 8)
    i = O.sparsity_info(rp, ci)
    assert i["bandwidth"] == 1 and i["is_banded"] and i["avg_nnz_per_row"] == 0.25 and i["sparsity_ratio"] == 2.0 / 64.0
    rp, ci, _ = O.csr_from_triplets([0], [2], [1.0], 8, 8)
    assert O.sparsity_info(rp, ci)["bandwidth"] == 2 and not O.sparsity_info(rp, ci)["is_banded"]      # 2 < 8 / 4 is false
