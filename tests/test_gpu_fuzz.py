"""Seeded random systems through the main entry points, bit for bit against the oracle: sizes that do not fill the last
slice / block, row lengths from 1 to several hundred in one matrix (batched, long-row and heavy-row paths), bandwidths on
both sides of every kernel-selection threshold (LDS band kernel with 4- and 8-wave blocks, general kernel), both summation
orders, sparse and dense push rounds, query sessions."""
import numpy as np
import pytest

import sublinear_time_solver_amd as S
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _bits_equal(a, b):
    return (np.ascontiguousarray(a).view(np.uint64) == np.ascontiguousarray(b).view(np.uint64)).all()


def _random_system(rng, n, w, max_len, long_rows):
    tr, tc, tv = [], [], []
    for i in range(n):
        m = int(rng.integers(0, max_len))
        if long_rows and rng.random() < 0.01:
            m = int(rng.integers(260, 700))
        lo, hi = (0, n) if w == 0 else (max(0, i - w), min(n, i + w + 1))
        m = min(m, hi - lo - 1)
        cols = rng.choice(np.arange(lo, hi), size=m, replace=False) if m > 0 else np.zeros(0, dtype=np.int64)
        cols = cols[cols != i]
        vals = rng.uniform(-1.0, 1.0, size=cols.size)
        tr += [i] * (cols.size + 1)
        tc += cols.tolist() + [i]
        tv += vals.tolist() + [float(rng.uniform(1.5, 3.0)) * np.abs(vals).sum() + 0.5]
    return O.csr_from_triplets(tr, tc, tv, n, n)


CASES = [(1, 0, 4, False), (63, 0, 9, False), (64, 5, 9, False), (65, 0, 20, False), (257, 40, 12, False), (1000, 0, 30, True),
         (2049, 300, 17, False), (3001, 1200, 24, True), (4097, 2000, 6, False), (5000, 4000, 40, True), (5003, 0, 3, False),
         (6000, 5500, 16, False), (700, 699, 60, True)]


@pytest.mark.parametrize("n,w,max_len,long_rows", CASES)
def test_random_systems_bitwise(gpu, n, w, max_len, long_rows):
    rng = np.random.default_rng(1000 * n + w)
    rp, ci, va = _random_system(rng, n, w, max_len, long_rows)
    m = S.SparseMatrix.from_csr(rp, ci, va, n, n, with_transpose=True)
    lens = np.diff(rp.astype(np.int64))
    dist = np.abs(ci.astype(np.int64) - np.repeat(np.arange(n), lens))
    info = m.info()
    assert 24 <= info.long_row_threshold <= 256 and info.n_long_rows == int((lens > info.long_row_threshold).sum())
    short = np.repeat(lens <= info.long_row_threshold, lens)        # longer rows live outside the slice layout
    assert info.bandwidth == int(dist[short].max(initial=0))
    x = rng.standard_normal(n)
    b = rng.standard_normal(n)
    bs = b * (rng.random(n) < 0.02)                                   # sparse right-hand side: local pushes
    if not bs.any():
        bs[0] = 1.0
    for order in (0, 1):
        assert _bits_equal(m.multiply_vector(x, order), O.spmv(rp, ci, va, x, order))
        g = S.NeumannSolver(order=order).solve(m, b, S.SolverOptions(tolerance=1e-12))
        o = O.neumann_solve(rp, ci, va, b, tolerance=1e-12, order=order)
        assert (g.iterations, g.converged) == (o["iterations"], o["converged"]) and _bits_equal(g.solution, o["x"])
        for ds in (2.0, 1.0 / 16.0, 1e-12):
            p = S.PushSolver(theta=1e-9, dense_switch=ds, order=order).solve(m, bs)
            q = O.push_sync_solve(rp, ci, va, bs, theta=1e-9, order=order)
            assert (p["rounds"], p["pushes"]) == (q["rounds"], q["pushes"]), (order, ds)
            assert _bits_equal(p["solution"], q["x"]) and _bits_equal(p["residual"], q["r"]), (order, ds)
    trp, tci, tva = O.csr_transpose(rp, ci, va, n)
    with S.QuerySession(m, b) as sess:
        for row in {0, n // 2, n - 1}:
            e = sess.estimate(row, theta=1e-7)
            unit = np.zeros(n)
            unit[row] = 1.0
            q = O.push_sync_solve(trp, tci, tva, unit, theta=1e-7)
            assert (e.rounds, e.pushes) == (q["rounds"], q["pushes"])
            est = float(np.dot(q["x"], b))
            assert abs(e.estimate - est) <= 1e-13 * max(1.0, np.abs(q["x"]).sum() * np.abs(b).max())


# ---- the paced column-panel layout under random structure --------------------------------------------------------------------
def _structured_system(rng, rows, cols, kind, max_len, dup):
    """rows x cols operator (cols >= rows, diagonal inside): columns uniform / banded / clustered around a few centres / a mix;
    empty rows, duplicates (the same column stored twice: kept, added twice in stored order), a few rows beyond the long-row limit"""
    tr, tc, tv = [], [], []
    centres = rng.integers(0, cols, size=5)
    for i in range(rows):
        m = int(rng.integers(0, max_len + 1))
        if rng.random() < 0.004:
            m = int(rng.integers(300, 900))
        m = min(m, cols - 1)
        k = kind if kind != "mix" else ("uniform", "band", "cluster")[i % 3]
        if k == "uniform":
            c = rng.integers(0, cols, size=m)
        elif k == "band":
            w = max(4, cols // 7)
            c = np.clip(i + rng.integers(-w, w + 1, size=m), 0, cols - 1)
        else:
            c = np.clip(centres[rng.integers(0, 5, size=m)] + rng.integers(-40, 41, size=m), 0, cols - 1)
        c = np.unique(c[c != i])
        if dup and c.size > 2 and i % 5 == 0:
            c = np.sort(np.concatenate([c, c[:2]]))                       # two duplicated columns
        v = rng.uniform(-1.0, 1.0, size=c.size)
        tr += [i] * (c.size + 1); tc += c.tolist() + [i]; tv += v.tolist() + [2.0 * np.abs(v).sum() + 1.0]
    return O.csr_from_triplets(tr, tc, tv, rows, cols)


PACED = [(70, 70, "uniform", 6, False, 1), (1000, 1000, "band", 20, True, 1), (4097, 4097, "cluster", 12, True, 2), (9001, 9001, "mix", 30, True, 3),
         (3000, 200_000, "uniform", 9, False, 2), (5000, 3_000_000, "cluster", 14, True, 1), (20_011, 20_011, "band", 16, False, 2),
         (12_345, 2_200_000, "mix", 25, True, 3), (257, 70_000, "uniform", 40, True, 1)]


@pytest.mark.parametrize("rows,cols,kind,max_len,dup,cus", PACED)
def test_paced_panels_random_structures_bitwise(gpu, monkeypatch, rows, cols, kind, max_len, dup, cus):
    """the paced layout forced on matrices it would never be chosen for: tiny and ragged tiles, one to three pretended CUs (several
    rounds), panels with one entry or hundreds, runs of any length, the same row on both sides of a panel boundary, empty rows,
    duplicates, rows left to the long-row kernel, rectangular operators whose super-panels are mostly empty"""
    monkeypatch.setenv("SL_PW_FORCE", "1")
    monkeypatch.setenv("SL_PW_CUS", str(cus))
    rng = np.random.default_rng(rows * 7 + cols + cus)
    rp, ci, va = _structured_system(rng, rows, cols, kind, max_len, dup)
    m = S.SparseMatrix.from_csr(rp, ci, va, rows, cols, column_panels=True)
    assert m.info().column_panels == 2
    for rep in range(2):
        x = rng.standard_normal(cols)
        assert _bits_equal(m.multiply_vector(x), O.spmv(rp, ci, va, x)), rep
    if rows == cols:
        b = rng.standard_normal(rows)
        g = S.NeumannSolver().solve(m, b, S.SolverOptions(tolerance=1e-11))
        o = O.neumann_solve(rp, ci, va, b, tolerance=1e-11)
        assert (g.iterations, g.converged) == (o["iterations"], o["converged"]) and _bits_equal(g.solution, o["x"])


# ---- XCD-local spans of the paced layout (sl_matrix::pw_xcd) ------------------------------------------------------------------------
@pytest.mark.parametrize("n,k,w,cus,groups,lo,hi", [(40_000, 16, 3_000, 4, 2, 0, 40_000), (50_003, 8, 9_000, 8, 4, 0, 50_003), (30_000, 5, 500, 8, 8, 0, 30_000),
                                                    (200_000, 16, 20_000, 4, 4, 60_000, 130_000), (66_000, 12, 30_000, 6, 3, 0, 66_000)])
def test_paced_panels_with_spans_dealt_inside_one_l2_bitwise(gpu, monkeypatch, n, k, w, cus, groups, lo, hi):
    """locality-bounded columns on the paced layout with a span of rows dealt among the blocks of ONE L2 (blocks b, b + G, ... are logical
    neighbours): forced on small banded systems on pretended devices of `cus` CUs in `groups` L2 groups, several rounds, also one rank's
    rows of a larger system — SpMV in the CSR order, the fused solve, the residual, dense push rounds: bit for bit"""
    from sublinear_time_solver_amd import generators as G
    monkeypatch.setenv("SL_PW_FORCE", "1")
    monkeypatch.setenv("SL_PW_CUS", str(cus))
    monkeypatch.setenv("SL_PW_XCD", str(groups))
    rp, ci, va, b = G.sdd_rows(n, k, 3, w, lo, hi)
    m = S.SparseMatrix.from_csr(rp, ci, va, hi - lo, n, row_offset=lo, column_panels=True, with_transpose=(lo == 0 and hi == n))
    assert m.info().column_panels == 2
    rng = np.random.default_rng(n + w)
    for rep in range(2):
        x = rng.standard_normal(n)
        assert _bits_equal(m.multiply_vector(x), O.spmv(rp, ci, va, x)), rep
    if lo == 0 and hi == n:
        g = S.NeumannSolver().solve(m, b, S.SolverOptions(tolerance=1e-11))
        o = O.neumann_solve(rp, ci, va, b, tolerance=1e-11)
        assert (g.iterations, g.converged) == (o["iterations"], o["converged"]) and _bits_equal(g.solution, o["x"])
        assert abs(g.residual_norm - o["residual_norm"]) <= 1e-12 * o["residual_norm"]
        bs = b * (np.arange(n) % 3 == 0)
        p = S.PushSolver(theta=1e-9, dense_switch=1e-12).solve(m, bs)
        q = O.push_sync_solve(rp, ci, va, bs, theta=1e-9)
        assert (p["rounds"], p["pushes"]) == (q["rounds"], q["pushes"]) and _bits_equal(p["solution"], q["x"]) and _bits_equal(p["residual"], q["r"])


def test_nothing_relies_on_fresh_device_memory_being_zero(gpu):
    """SL_POISON_ALLOC=1 fills every device allocation of the library (and every block its workspace cache hands out again) with 0xA5
    bytes before use; the parity and fuzz files must pass under it exactly as they do on an idle box's zero pages"""
    import os
    import subprocess
    import sys
    if os.environ.get("SL_POISON_ALLOC") == "1":
        pytest.skip("already inside the child run")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_fuzz.py"), os.path.join(here, "test_gpu_parity.py"),
                        os.path.join(here, "test_gpu_panels.py"), "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, timeout=1500, env=dict(os.environ, SL_POISON_ALLOC="1"))
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
