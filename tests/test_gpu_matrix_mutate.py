"""The two `&mut self` methods of the matrix — SparseMatrix::scale / add_diagonal (matrix/mod.rs:346-372 over CSRStorage::scale /
add_diagonal, matrix/sparse.rs:229-248) — and solver::utils beyond l2_norm (solver/mod.rs:374-461), SolverOptions::streaming
(:101-116), SolverResult::meets_quality_criteria (:192-195): the device against the oracle's restatement.

A mutator must leave EVERY layout copy of the matrix consistent: the row slices (sl_spmv, sl_matrix_get / row), the raw CSR
(download), the transpose (the push), hub rows (raw CSR entries), and the sorted column streams (paced panels, forced here on a small
system).  Each is read back through the path that uses it and compared bit for bit with the oracle applied to the plain CSR."""
import ctypes as C
import os

import numpy as np
import pytest

import sublinear_time_solver_amd as S
from sublinear_time_solver_amd import _lib as L
from sublinear_time_solver_amd import generators as G
from sublinear_time_solver_amd import utils as U
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def _ragged(n, seed, with_hub=True, drop_diag_every=0, dup_diag_rows=()):
    """a ragged diagonally dominant CSR: rows of 1..9 entries, one hub row, optionally rows WITHOUT a diagonal entry and rows that store
    their diagonal twice (duplicates are separate entries, sparse.rs:80-132)"""
    rng = np.random.default_rng(seed)
    tr, tc, tv = [], [], []
    for i in range(n):
        k = int(rng.integers(0, 9))
        cols = set(int(c) for c in rng.integers(0, n, size=k)) - {i}
        if with_hub and i == n // 3:
            cols = set(int(c) for c in rng.choice(n, size=min(n - 1, 700), replace=False)) - {i}
        for c in sorted(cols):
            tr.append(i), tc.append(c), tv.append(float(rng.uniform(-1.0, 1.0)))
        if not (drop_diag_every and i % drop_diag_every == 2):
            tr.append(i), tc.append(i), tv.append(float(len(cols)) + 1.5 + float(rng.uniform(0.0, 1.0)))
            if i in dup_diag_rows:
                tr.append(i), tc.append(i), tv.append(0.25)
    return O.csr_from_triplets(tr, tc, tv, n, n)


def _download(m, n, nnz):
    rp, ci, va = np.zeros(n + 1, dtype=np.uint32), np.zeros(nnz, dtype=np.uint32), np.zeros(nnz)
    L.check(L.load().sl_matrix_download_csr(m._h, L.ptr(rp), L.ptr(ci), L.ptr(va)))
    return rp, ci, va


def _rows(m, n):
    """every row through Matrix::row_iter (the slice layout / the hub rows' raw entries)"""
    cols, vals = [], []
    for i in range(n):
        p = list(m.row_iter(i))
        cols += [c for c, _ in p]
        vals += [v for _, v in p]
    return np.asarray(cols, dtype=np.uint32), np.asarray(vals, dtype=np.float64)


@pytest.mark.parametrize("factor", [-2.5, 1.0 / 3.0, 0.0])
def test_scale_every_layout(gpu, factor):
    n = 700
    rp, ci, va = _ragged(n, seed=5)
    m = S.SparseMatrix.from_csr(rp, ci, va, n, n, with_transpose=True, keep_csr=True)
    assert m.info().n_long_rows >= 1
    m.scale(factor)
    want = O.csr_scale(va, factor)
    _, _, got = _download(m, n, va.size)
    assert (_bits(got) == _bits(want)).all(), "raw CSR values after scale"
    rc, rv = _rows(m, n)
    assert (rc == ci).all() and (_bits(rv) == _bits(want)).all(), "row slices / hub rows after scale"
    x = np.random.default_rng(1).standard_normal(n)
    assert (_bits(m.multiply_vector(x)) == _bits(O.spmv(rp, ci, want, x))).all(), "SpMV after scale"
    # the transpose: A^T as a matrix of its own, its rows against the oracle's transpose of the scaled CSR
    mt = m.transpose(keep_csr=True)
    trp, tci, tva = O.csr_transpose(rp, ci, want, n)
    _, tci_d, tva_d = _download(mt, n, va.size)
    assert (tci_d == tci).all() and (_bits(tva_d) == _bits(tva)).all(), "transpose after scale"


def test_add_diagonal_skips_rows_without_one_and_takes_the_searched_duplicate(gpu):
    n = 700
    rp, ci, va = _ragged(n, seed=9, drop_diag_every=7, dup_diag_rows=(0, 5, 699, 233))
    m = S.SparseMatrix.from_csr(rp, ci, va, n, n, with_transpose=True, keep_csr=True)
    alpha = 0.375
    want, changed = O.csr_add_diagonal(rp, ci, va, alpha)
    assert 0 < changed < n                                             # rows without a diagonal entry exist and are skipped
    m.add_diagonal(alpha)
    _, _, got = _download(m, n, va.size)
    assert (_bits(got) == _bits(want)).all(), "raw CSR values after add_diagonal"
    assert int((got != va).sum()) == changed                          # ONE entry per row changed, also where the diagonal is stored twice
    rc, rv = _rows(m, n)
    assert (rc == ci).all() and (_bits(rv) == _bits(want)).all(), "row slices / hub rows after add_diagonal"
    for i in (0, 2, 5, 233, 699):                                      # Matrix::get lands on the same duplicate the mutator changed
        assert m.get(i, i) == O.matrix_get(rp, ci, want, i, i)
    x = np.random.default_rng(2).standard_normal(n)
    assert (_bits(m.multiply_vector(x)) == _bits(O.spmv(rp, ci, want, x))).all()
    mt = m.transpose(keep_csr=True)
    trp, tci, tva = O.csr_transpose(rp, ci, want, n)
    _, tci_d, tva_d = _download(mt, n, va.size)
    assert (tci_d == tci).all() and (_bits(tva_d) == _bits(tva)).all(), "transpose after add_diagonal"


def test_mutators_on_a_matrix_that_keeps_only_its_row_slices(gpu):
    """no raw CSR, no transpose: the slice layout is the only copy"""
    n, k = 3000 + 7, 8
    rp, ci, va, b = G.sdd_rows(n, k, seed=4, half_bandwidth=40)
    m = S.SparseMatrix.from_csr(rp, ci, va, n, n)
    m.scale(0.5)
    m.add_diagonal(-0.125)
    want, changed = O.csr_add_diagonal(rp, ci, O.csr_scale(va, 0.5), -0.125)
    assert changed == n
    x = np.random.default_rng(3).standard_normal(n)
    assert (_bits(m.multiply_vector(x)) == _bits(O.spmv(rp, ci, want, x))).all()
    assert (_bits(m.multiply_vector(x, order=L.SL_ORDER_SIMD4)) == _bits(O.spmv(rp, ci, want, x, order=O.ORDER_SIMD4))).all()
    grp, gci, gva = m.to_csr()                                           # no raw copy kept: as_csr writes the rows back from the slices
    assert (grp == rp).all() and (gci == ci).all() and (_bits(gva) == _bits(want)).all()
    assert m.to_triplets()[:3] == [(0, int(ci[0]), float(want[0])), (0, int(ci[1]), float(want[1])), (0, int(ci[2]), float(want[2]))]
    # a shifted system solves like the re-uploaded one, bit for bit (what the mutators are for)
    g = S.NeumannSolver().solve(m, b, S.SolverOptions(tolerance=1e-10))
    o = O.neumann_solve(rp, ci, want, b, tolerance=1e-10)
    assert g.iterations == o["iterations"] and (_bits(g.solution) == _bits(o["x"])).all()


@pytest.mark.parametrize("op", ["scale", "scale_nonfinite", "add_diagonal"])
def test_mutators_rebuild_the_sorted_column_streams(gpu, op):
    """paced column panels forced on a small uniform-column system (as smoke() does): after a mutation the headline kernel must read the
    NEW values — in place for a finite scale, rebuilt from the updated rows otherwise"""
    os.environ["SL_PW_FORCE"], os.environ["SL_PW_CUS"] = "1", "2"
    try:
        n = 20_000 + 3
        rp, ci, va, b = G.sdd_rows(n, 16, seed=3)
        m = S.SparseMatrix.from_csr(rp, ci, va, n, n, column_panels=True)
        assert m.info().column_panels == 2
        bytes_before = m.info().device_bytes
        if op == "scale":
            m.scale(1.25)
            want = O.csr_scale(va, 1.25)
        elif op == "scale_nonfinite":
            m.scale(float("inf"))
            want = O.csr_scale(va, float("inf"))
        else:
            m.add_diagonal(2.0)
            want, _ = O.csr_add_diagonal(rp, ci, va, 2.0)
        assert m.info().column_panels == 2 and m.info().device_bytes == bytes_before
        x = np.random.default_rng(5).standard_normal(n)
        y, yo = m.multiply_vector(x), O.spmv(rp, ci, want, x)
        assert (_bits(y) == _bits(yo)).all() or (op == "scale_nonfinite" and (np.isnan(y) == np.isnan(yo)).all() and (_bits(y)[~np.isnan(y)] == _bits(yo)[~np.isnan(yo)]).all())
        if op != "scale_nonfinite":
            g = S.NeumannSolver().solve(m, b, S.SolverOptions(tolerance=1e-10, max_iterations=200))
            o = O.neumann_solve(rp, ci, want, b, tolerance=1e-10, max_iterations=200)
            assert g.iterations == o["iterations"] and (_bits(g.solution) == _bits(o["x"])).all(), "paced panel kernel after the mutation"
    finally:
        os.environ.pop("SL_PW_FORCE", None); os.environ.pop("SL_PW_CUS", None)


def test_add_diagonal_refuses_non_square_and_accepts_row_slices(gpu):
    wide = S.SparseMatrix.from_triplets([(0, 0, 2.0), (1, 1, 3.0), (1, 2, 1.0)], 2, 3)
    with pytest.raises(S.SolverError) as e:
        wide.add_diagonal(1.0)
    assert e.value.kind == "InvalidInput" and "non-square" in str(e.value)      # matrix/mod.rs:356-361
    wide.scale(2.0)                                                             # scale has no such rule
    assert wide.get(1, 2) == 2.0
    # rows [4, 9) of a 12 x 12 system: the own column of local row r is 4 + r
    n, lo, hi = 12, 4, 9
    rp, ci, va = _ragged(n, seed=11, with_hub=False)
    sl_rp = (rp[lo:hi + 1] - rp[lo]).astype(np.uint32)
    sl_ci, sl_va = ci[rp[lo]:rp[hi]], va[rp[lo]:rp[hi]]
    m = S.SparseMatrix.from_csr(sl_rp, sl_ci, sl_va, hi - lo, n, row_offset=lo, keep_csr=True)
    m.add_diagonal(0.5)
    want, changed = O.csr_add_diagonal(sl_rp, sl_ci, sl_va, 0.5, row_offset=lo)
    assert changed == hi - lo
    _, _, got = _download(m, hi - lo, sl_va.size)
    assert (_bits(got) == _bits(want)).all()
    # the range that starts at row 0 looks like a wide matrix: SL_MATRIX_ROW_SLICE says what it is
    first = S.SparseMatrix.from_csr(rp[:lo + 1], ci[:rp[lo]], va[:rp[lo]], lo, n, keep_csr=True)
    with pytest.raises(S.SolverError):
        first.add_diagonal(0.5)
    first = S.SparseMatrix.from_csr(rp[:lo + 1], ci[:rp[lo]], va[:rp[lo]], lo, n, keep_csr=True, row_slice=True)
    first.add_diagonal(0.5)
    want0, changed0 = O.csr_add_diagonal(rp[:lo + 1], ci[:rp[lo]], va[:rp[lo]], 0.5)
    _, _, got0 = _download(first, lo, int(rp[lo]))
    assert changed0 == lo and (_bits(got0) == _bits(want0)).all()


def test_solver_utils_norms_residual_convergence(gpu):
    rng = np.random.default_rng(8)
    v = rng.standard_normal(100_003) * 10.0 ** rng.integers(-3, 4, size=100_003)
    assert U.linf_norm(v) == O.linf_norm(v)                                     # a maximum: exact
    assert abs(U.l1_norm(v) - O.l1_norm(v)) <= 1e-12 * O.l1_norm(v)
    assert abs(U.l2_norm(v) - O.l2_norm(v)) <= 1e-12 * O.l2_norm(v)
    w = v.copy(); w[17] = np.nan; w[5] = -1e9
    assert U.linf_norm(w) == 1e9 == O.linf_norm(w)                              # f64::max skips NaN (solver/mod.rs:379-381)
    assert U.linf_norm(np.zeros(0)) == 0.0 and U.l1_norm(np.zeros(0)) == 0.0
    for t, ref in (("l1", O.l1_norm(v)), ("l2", O.l2_norm(v)), ("linf", O.linf_norm(v)), ("weighted", O.l2_norm(v))):   # Weighted -> L2, :389
        assert abs(U.compute_norm(v, t) - ref) <= 1e-12 * ref
    n, k = 5000, 8
    rp, ci, va, b = G.sdd_rows(n, k, seed=6)
    m = S.SparseMatrix.from_csr(rp, ci, va, n, n)
    x = rng.standard_normal(n)
    assert (_bits(U.compute_residual(m, x, b)) == _bits(O.spmv(rp, ci, va, x) - b)).all()       # r = A x, then r -= b (solver/mod.rs:394-405)
    # check_convergence, all five modes (solver/mod.rs:408-461)
    cur, prev = x, x + 1e-7 * rng.standard_normal(n)
    change = float(np.linalg.norm(cur - prev))
    assert U.check_convergence(1e-7, 1e-6, "residual_norm", 0.0, None, cur) and not U.check_convergence(1e-5, 1e-6, "residual_norm", 0.0, None, cur)
    assert U.check_convergence(1e-3, 1e-6, "relative_residual", 1e4, None, cur) and not U.check_convergence(1e-3, 1e-6, "relative_residual", 0.0, None, cur)
    assert not U.check_convergence(0.0, 1.0, "solution_change", 1.0, None, cur)                # no previous solution: false
    assert U.check_convergence(0.0, change * 1.01, "solution_change", 1.0, prev, cur) and not U.check_convergence(0.0, change * 0.99, "solution_change", 1.0, prev, cur)
    rel = change / float(np.linalg.norm(prev))
    assert U.check_convergence(0.0, rel * 1.01, "relative_solution_change", 1.0, prev, cur) and not U.check_convergence(0.0, rel * 0.99, "relative_solution_change", 1.0, prev, cur)
    assert U.check_convergence(0.0, 0.0, "relative_solution_change", 1.0, np.zeros(4), np.zeros(4))     # ||prev|| = 0: the absolute test
    assert U.check_convergence(1e-7, 1e-6, "combined", 0.0, None, cur) and not U.check_convergence(1e-7, 1e-6, "combined", 1e-3, None, cur)


def test_streaming_preset_and_quality_criteria(gpu):
    o = L.NeumannOptions()
    d = L.NeumannOptions()
    lib = L.load()
    lib.sl_neumann_options_streaming(C.byref(o))
    lib.sl_neumann_options_default(C.byref(d))
    assert (o.tolerance, o.max_iterations, o.collect_stats, o.compute_error_bounds) == (1e-4, 1000, 1, 0)     # solver/mod.rs:101-116
    assert (o.max_terms, o.series_tolerance, o.order, o.start, o.residual, o.mem) == (d.max_terms, d.series_tolerance, d.order, d.start, d.residual, d.mem)
    so = S.SolverOptions.streaming(25)
    assert (so.tolerance, so.max_iterations, so.collect_stats, so.compute_error_bounds, so.streaming_interval) == (1e-4, 1000, True, False, 25)
    n = 2000
    rp, ci, va, b = G.sdd_rows(n, 8, seed=2)
    m = S.SparseMatrix.from_csr(rp, ci, va, n, n)
    r = S.NeumannSolver().solve(m, b, so)
    assert r.converged and r.meets_quality_criteria(1e-4) and not r.meets_quality_criteria(r.residual_norm * 0.5)  # solver/mod.rs:192-195
    res = L.NeumannResult()
    res.converged, res.residual_norm = 1, 1e-5
    assert lib.sl_neumann_result_meets_quality_criteria(C.byref(res), 1e-4) == 1 and lib.sl_neumann_result_meets_quality_criteria(C.byref(res), 1e-6) == 0
    res.converged = 0
    assert lib.sl_neumann_result_meets_quality_criteria(C.byref(res), 1e-4) == 0
