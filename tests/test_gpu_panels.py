"""Column-panel layout (sl_panel_kernel): the entries regrouped by (tile of rows, panel of columns) and summed through running
sums in LDS must give every row the bits of the sequential reference loop — uniform columns, ragged rows with hubs and duplicate
entries, a row slice of a larger system, every epilogue (SpMV, fused Neumann step, residual, dense push round)."""
import numpy as np
import pytest

import sublinear_time_solver_amd as S
from sublinear_time_solver_amd import _lib as L
from sublinear_time_solver_amd import generators as G
from oracle import oracle as O

pytestmark = pytest.mark.gpu


# the layout small matrices get: dynamic tiles (1) — or the paced layout (2) when a run forces it on every size (SL_PW_FORCE=1:
# the whole file then exercises the paced kernel on these inputs as well)
import os
_SMALL = 2 if os.environ.get("SL_PW_FORCE") == "1" else 1


def _bits_equal(a, b):
    return (np.ascontiguousarray(a).view(np.uint64) == np.ascontiguousarray(b).view(np.uint64)).all()


@pytest.mark.parametrize("n,k", [(300_000, 16), (70_001, 5), (5000, 8)])
def test_uniform_columns_spmv_neumann_residual(gpu, n, k):
    rp, ci, va, b = G.sdd_rows(n, k, seed=4)                               # w = 0: columns all over the vector
    mp = S.SparseMatrix.from_csr(rp, ci, va, n, n, column_panels=True)
    mg = S.SparseMatrix.from_csr(rp, ci, va, n, n, column_panels=False)
    assert mp.info().column_panels == _SMALL and mg.info().column_panels == 0
    assert mp.info().device_bytes > mg.info().device_bytes
    x = np.cos(np.arange(n) * 0.11) + 0.3
    ref = O.spmv(rp, ci, va, x)
    assert _bits_equal(mp.multiply_vector(x), ref) and _bits_equal(mg.multiply_vector(x), ref)
    o = O.neumann_solve(rp, ci, va, b, tolerance=1e-11)
    for m in (mp, mg):
        g = S.NeumannSolver().solve(m, b, S.SolverOptions(tolerance=1e-11))
        assert g.converged and g.iterations == o["iterations"] and _bits_equal(g.solution, o["x"])
        np.testing.assert_allclose(g.term_norms, o["term_norms"], rtol=1e-10)      # the CPU sum runs sequentially over n terms
        assert abs(g.residual_norm - o["residual_norm"]) <= 1e-10 * max(1.0, o["residual_norm"])
    # the 4-lane order is not served by the panels: same answer as without them
    g4 = S.NeumannSolver(order=L.SL_ORDER_SIMD4).solve(mp, b, S.SolverOptions(tolerance=1e-11))
    o4 = O.neumann_solve(rp, ci, va, b, tolerance=1e-11, order=O.ORDER_SIMD4)
    assert _bits_equal(g4.solution, o4["x"])


def _ragged_system(n=9000, seed=3):
    """row dominant: 2..30 entries per row, hubs of 400..3000, a few rows with the same column stored twice"""
    rng = np.random.default_rng(seed)
    tr, tc, tv = [], [], []
    for i in range(n):
        m = int(rng.integers(2, 31))
        if i % 211 == 0:
            m = int(rng.integers(400, 3000))
        cols = np.sort(rng.choice(n - 1, size=m - 1, replace=False))
        cols = cols + (cols >= i)
        if i % 17 == 0 and cols.size > 2:
            cols[1] = cols[0]                                              # duplicate entry: kept, added twice in stored order
        vals = rng.uniform(-1.0, 1.0, size=cols.size)
        tr += [i] * (cols.size + 1); tc += cols.tolist() + [i]; tv += vals.tolist() + [2.0 * np.abs(vals).sum() + 1.0]
    return O.csr_from_triplets(tr, tc, tv, n, n)


def test_ragged_rows_hubs_duplicates_and_dense_push_rounds(gpu):
    rp, ci, va = _ragged_system()
    n = rp.size - 1
    mp = S.SparseMatrix.from_csr(rp, ci, va, n, n, with_transpose=True, column_panels=True)
    assert mp.info().column_panels == _SMALL and mp.info().n_long_rows > 0
    x = np.sin(np.arange(n) * 0.7) - 0.2
    assert _bits_equal(mp.multiply_vector(x), O.spmv(rp, ci, va, x))
    b = 1.0 + (np.arange(n) % 7) * 0.5
    o = O.neumann_solve(rp, ci, va, b, tolerance=1e-10)
    g = S.NeumannSolver().solve(mp, b, S.SolverOptions(tolerance=1e-10))
    assert g.iterations == o["iterations"] and _bits_equal(g.solution, o["x"])
    # thresholded push with dense rounds forced from the start (dense_switch tiny): the PUSH epilogue of the panel kernel
    q = O.push_sync_solve(rp, ci, va, b, theta=1e-8, log_cap=1 << 22)
    p = S.PushSolver(theta=1e-8, dense_switch=1e-9).solve(mp, b, log_frontier=1 << 22)
    assert p["converged"] and p["rounds"] == q["rounds"] and p["dense_rounds"] > 0
    assert (p["frontier_log"] == q["frontier_log"]).all()
    assert _bits_equal(p["solution"], q["x"]) and _bits_equal(p["residual"], q["r"])


def test_row_slice_of_a_larger_system(gpu):
    """rows [lo, hi) of a system as their own matrix with global column ids (what a rank of a partitioned solve holds)"""
    n, k, lo, hi = 120_000, 12, 33_333, 91_777
    rp, ci, va, b = G.sdd_rows(n, k, seed=6)
    prp = (rp[lo:hi + 1].astype(np.int64) - int(rp[lo])).astype(np.uint32)
    pci, pva = ci[rp[lo]:rp[hi]], va[rp[lo]:rp[hi]]
    m = S.SparseMatrix.from_csr(prp, pci, pva, hi - lo, n, row_offset=lo, column_panels=True)
    assert m.info().column_panels == _SMALL
    x = np.cos(np.arange(n) * 0.05)
    assert _bits_equal(m.multiply_vector(x), O.spmv(rp, ci, va, x)[lo:hi])


def test_thin_panels_same_row_twice_in_one_chunk(gpu):
    """few entries per (tile, panel): 64 consecutive entries of a tile's stream then span several panels and can hold the same row
    more than once, far apart — the parts must be applied one after the other (a rectangular operator: 5000 rows, 6 million columns,
    46 panels, one to three entries per row)"""
    rng = np.random.default_rng(12)
    rows, cols = 5000, 6_000_000
    cnt = rng.integers(1, 4, size=rows)
    rp = np.zeros(rows + 1, dtype=np.uint32)
    rp[1:] = np.cumsum(cnt)
    ci = np.concatenate([np.sort(rng.choice(cols, size=int(c), replace=False)) for c in cnt]).astype(np.uint32)
    va = rng.uniform(-1.0, 1.0, size=ci.size)
    m = S.SparseMatrix.from_csr(rp, ci, va, rows, cols, column_panels=True)
    assert m.info().column_panels == _SMALL
    x = rng.uniform(-1.0, 1.0, size=cols)
    assert _bits_equal(m.multiply_vector(x), O.spmv(rp, ci, va, x))


def test_seven_million_short_rows_many_thin_panels(gpu):
    """n = 7.1 million rows of 1..4 off-diagonal entries with uniformly random columns: 109 panels, about 50 entries per
    (tile, panel) group — every chunk of a tile's stream spans several panels; duplicates and a few hub rows included.
    SpMV, four fused Neumann steps and a full solve, bit for bit against the CPU restatement."""
    rng = np.random.default_rng(77)
    n = 7_100_003
    cnt = rng.integers(1, 5, size=n)
    cnt[rng.choice(n, size=40, replace=False)] = rng.integers(300, 2000, size=40)        # hubs: long rows
    rows = np.repeat(np.arange(n, dtype=np.int64), cnt)
    cols = rng.integers(0, n, size=rows.size, dtype=np.int64)
    cols = np.where(cols == rows, (cols + 1) % n, cols)
    vals = rng.uniform(-1.0, 1.0, size=rows.size)
    off = np.zeros(n)
    np.add.at(off, rows, np.abs(vals))
    rows = np.concatenate([rows, np.arange(n)]); cols = np.concatenate([cols, np.arange(n)]); vals = np.concatenate([vals, 2.0 * off + 1.0])
    order = np.lexsort((cols, rows))                                                      # stable: duplicate (row, col) keep their order
    rows, cols, vals = rows[order], cols[order], vals[order]
    rp = np.zeros(n + 1, dtype=np.uint32)
    rp[1:] = np.cumsum(np.bincount(rows, minlength=n))
    ci, va = cols.astype(np.uint32), vals
    m = S.SparseMatrix.from_csr(rp, ci, va, n, n, column_panels=True)
    info = m.info()
    assert info.column_panels == 2 and info.n_long_rows >= 40          # balanced tiles at this size: the paced persistent layout
    x = np.cos(np.arange(n) * 0.001) + 0.2
    assert _bits_equal(m.multiply_vector(x), O.spmv(rp, ci, va, x))
    b = 1.0 + (np.arange(n) % 11) * 0.1
    o = O.neumann_solve(rp, ci, va, b, tolerance=1e-9)
    g = S.NeumannSolver().solve(m, b, S.SolverOptions(tolerance=1e-9))
    assert g.converged and g.iterations == o["iterations"] and _bits_equal(g.solution, o["x"])


# ---- the paced layout (sl_pw_kernel): persistent 16-wave blocks, tiles dealt in rounds -----------------------------------------
@pytest.fixture
def paced(monkeypatch):
    """build the paced layout whatever the size / balance, on a pretended 2-CU device: 32 waves, so small systems take several rounds"""
    monkeypatch.setenv("SL_PW_FORCE", "1")
    monkeypatch.setenv("SL_PW_CUS", "2")


@pytest.mark.parametrize("n,k", [(100_000, 16), (70_001, 5), (3000, 8)])
def test_paced_uniform_columns_all_epilogues(gpu, paced, n, k):
    rp, ci, va, b = G.sdd_rows(n, k, seed=9)
    m = S.SparseMatrix.from_csr(rp, ci, va, n, n, with_transpose=True, column_panels=True)
    assert m.info().column_panels == 2
    x = np.cos(np.arange(n) * 0.13) - 0.4
    assert _bits_equal(m.multiply_vector(x), O.spmv(rp, ci, va, x))
    o = O.neumann_solve(rp, ci, va, b, tolerance=1e-11)
    g = S.NeumannSolver().solve(m, b, S.SolverOptions(tolerance=1e-11))
    assert g.converged and g.iterations == o["iterations"] and _bits_equal(g.solution, o["x"])
    np.testing.assert_allclose(g.term_norms, o["term_norms"], rtol=1e-10)
    assert abs(g.residual_norm - o["residual_norm"]) <= 1e-10 * max(1.0, o["residual_norm"])
    bs = b * (np.arange(n) % 3 == 0)
    q = O.push_sync_solve(rp, ci, va, bs, theta=1e-8, log_cap=1 << 22)
    p = S.PushSolver(theta=1e-8, dense_switch=1e-9).solve(m, bs, log_frontier=1 << 22)       # dense rounds: the PUSH epilogue
    assert p["converged"] and p["rounds"] == q["rounds"] and p["dense_rounds"] > 0
    assert (p["frontier_log"] == q["frontier_log"]).all() and _bits_equal(p["solution"], q["x"]) and _bits_equal(p["residual"], q["r"])


def test_paced_ragged_rows_hubs_duplicates(gpu, paced):
    """unequal tiles (the pace then waits for the slowest wave), runs of one row inside a panel (duplicates; rows of up to 30 entries
    over a 9000-column vector = ONE panel: every row is a run, many longer than four), long rows left to the long-row kernel"""
    rp, ci, va = _ragged_system()
    n = rp.size - 1
    m = S.SparseMatrix.from_csr(rp, ci, va, n, n, column_panels=True)
    assert m.info().column_panels == 2 and m.info().n_long_rows > 0
    x = np.sin(np.arange(n) * 0.7) - 0.2
    assert _bits_equal(m.multiply_vector(x), O.spmv(rp, ci, va, x))
    b = 1.0 + (np.arange(n) % 7) * 0.5
    o = O.neumann_solve(rp, ci, va, b, tolerance=1e-10)
    g = S.NeumannSolver().solve(m, b, S.SolverOptions(tolerance=1e-10))
    assert g.iterations == o["iterations"] and _bits_equal(g.solution, o["x"])


def test_paced_thin_panels_and_empty_super_panels(gpu, paced):
    """5000 rows x 9 million columns, one to three entries per row, columns only in [0, 10^6) and [7 * 10^6, 9 * 10^6): 64 consecutive
    entries of a tile's stream span many panels (the same row twice, far apart), and whole super-panels of 2^20 columns are empty —
    the stream bridges them with padding entries"""
    rng = np.random.default_rng(21)
    rows, cols = 5000, 9_000_000
    cnt = rng.integers(1, 4, size=rows)
    rp = np.zeros(rows + 1, dtype=np.uint32)
    rp[1:] = np.cumsum(cnt)
    pool = np.concatenate([np.arange(0, 1_000_000), np.arange(7_000_000, 9_000_000)])
    ci = np.concatenate([np.sort(rng.choice(pool, size=int(c), replace=False)) for c in cnt]).astype(np.uint32)
    va = rng.uniform(-1.0, 1.0, size=ci.size)
    m = S.SparseMatrix.from_csr(rp, ci, va, rows, cols, column_panels=True)
    assert m.info().column_panels == 2
    x = rng.uniform(-1.0, 1.0, size=cols)
    assert _bits_equal(m.multiply_vector(x), O.spmv(rp, ci, va, x))


def test_paced_row_slice(gpu, paced):
    n, k, lo, hi = 120_000, 12, 33_333, 91_777
    rp, ci, va, b = G.sdd_rows(n, k, seed=6)
    prp = (rp[lo:hi + 1].astype(np.int64) - int(rp[lo])).astype(np.uint32)
    pci, pva = ci[rp[lo]:rp[hi]], va[rp[lo]:rp[hi]]
    m = S.SparseMatrix.from_csr(prp, pci, pva, hi - lo, n, row_offset=lo, column_panels=True)
    assert m.info().column_panels == 2
    x = np.cos(np.arange(n) * 0.05)
    assert _bits_equal(m.multiply_vector(x), O.spmv(rp, ci, va, x)[lo:hi])


def test_paced_runs_longer_than_four_ending_a_panel_followed_by_the_same_row(gpu, monkeypatch):
    """A band of half-width 14000 over 120 000 columns through the paced layout with the device's real CU count: tiles of 16 rows,
    every row a run of up to 16 entries inside one panel — and the rows around column 65536 with 15 entries in panel 0 and one in
    panel 1: the row that ends a tile's panel-0 segment is the row that starts its panel-1 segment, inside the same 64 entries.  The
    first panel's run (longer than four) has to be complete before the next panel's entry is added (a regression test: the tail of
    long runs used to be applied after ALL panels' heads)."""
    monkeypatch.setenv("SL_PW_FORCE", "1")
    n, k, w = 120_000, 16, 14_000
    rp, ci, va, b = G.sdd_rows(n, k, seed=9, half_bandwidth=w)
    m = S.SparseMatrix.from_csr(rp, ci, va, n, n, column_panels=True)
    assert m.info().column_panels == 2
    x = np.sin(np.arange(n) * 0.05) + 0.3
    assert _bits_equal(m.multiply_vector(x), O.spmv(rp, ci, va, x))
    lo, hi = 41_111, 99_999
    prp = (rp[lo:hi + 1].astype(np.int64) - int(rp[lo])).astype(np.uint32)
    ms = S.SparseMatrix.from_csr(prp, ci[rp[lo]:rp[hi]], va[rp[lo]:rp[hi]], hi - lo, n, row_offset=lo, column_panels=True)
    assert ms.info().column_panels == 2 and _bits_equal(ms.multiply_vector(x), O.spmv(rp, ci, va, x)[lo:hi])
    o = O.neumann_solve(rp, ci, va, b, tolerance=1e-10)
    g = S.NeumannSolver().solve(m, b, S.SolverOptions(tolerance=1e-10))
    assert g.iterations == o["iterations"] and _bits_equal(g.solution, o["x"])
