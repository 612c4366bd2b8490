#!/usr/bin/env python3
"""A timed random campaign over the same generators as tests/test_gpu_fuzz.py with FRESH seeds and random parameters — the fixed-seed
tests guard against regressions, this looks for what they do not cover.  Every case is checked bit for bit against the oracle (counts
and lists included); the first line of a failure carries everything needed to replay it (`--replay "<kind> <seed>"`).

usage: python tests/fuzz_campaign.py --seconds 300 [--kinds general,paced,orderany,acl,southwell,cg,wideband,session,trait,walk,mutate,walkserial] [--seed0 S]
Prints one JSON line: cases per kind, failures (each with kind + seed).  Exit status 1 when anything failed."""
import argparse
import json
import os
import sys
import time
import traceback

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import sublinear_time_solver_amd as S                      # noqa: E402
from sublinear_time_solver_amd import _lib as L            # noqa: E402
from oracle import oracle as O                             # noqa: E402
from tests import test_gpu_fuzz as F                       # noqa: E402

EPS = np.finfo(np.float64).eps
SEEN = {}                                                  # (kind, layout reported by sl_matrix_get_info) -> cases: what the campaign really exercised


def seen(kind, m):
    i = m.info()
    key = f"{kind}: column_panels={i.column_panels} long_rows={'yes' if i.n_long_rows else 'no'}"
    SEEN[key] = SEEN.get(key, 0) + 1


def bits_equal(a, b):
    return bool((np.ascontiguousarray(a, dtype=np.float64).view(np.uint64) == np.ascontiguousarray(b, dtype=np.float64).view(np.uint64)).all())


def case_general(seed):
    """any layout the library picks by itself: SpMV in both orders, full solve, synchronous push at three dense-switch settings"""
    rng = np.random.default_rng(seed)
    n = int(rng.choice([1, 2, 63, 64, 65, 130, 1000, 2500, 4099, 7000]))
    w = int(rng.choice([0, 0, 1, 7, 300, 1500, 4000, 6500])) if n > 8 else 0
    w = min(w, n - 1)
    max_len = int(rng.integers(1, 48))
    long_rows = bool(rng.random() < 0.4) and n >= 700
    rp, ci, va = F._random_system(rng, n, w, max_len, long_rows)
    m = S.SparseMatrix.from_csr(rp, ci, va, n, n, with_transpose=True)
    seen("general", m)
    x = rng.standard_normal(n)
    b = rng.standard_normal(n)
    bs = b * (rng.random(n) < 0.03)
    if not bs.any():
        bs[0] = 1.0
    tol = float(rng.choice([1e-6, 1e-10, 1e-13]))
    theta = float(rng.choice([1e-5, 1e-9]))
    for order in (0, 1):
        assert bits_equal(m.multiply_vector(x, order), O.spmv(rp, ci, va, x, order)), ("spmv", order)
        g = S.NeumannSolver(order=order).solve(m, b, S.SolverOptions(tolerance=tol))
        o = O.neumann_solve(rp, ci, va, b, tolerance=tol, order=order)
        assert (g.iterations, bool(g.converged)) == (o["iterations"], bool(o["converged"])) and bits_equal(g.solution, o["x"]), ("neumann", order)
        # norms are fixed-order tree sums on the device, sequential in the reference: equal to rounding, not to the bit
        assert abs(g.residual_norm - o["residual_norm"]) <= 1e-12 * max(o["residual_norm"], 1e-300), ("residual", order)
        ds = float(rng.choice([2.0, 1.0 / 16.0, 1e-12]))
        p = S.PushSolver(theta=theta, dense_switch=ds, order=order).solve(m, bs)
        q = O.push_sync_solve(rp, ci, va, bs, theta=theta, order=order)
        assert (p["rounds"], p["pushes"]) == (q["rounds"], q["pushes"]), ("push counts", order, ds)
        assert bits_equal(p["solution"], q["x"]) and bits_equal(p["residual"], q["r"]), ("push", order, ds)


def case_paced(seed):
    """the paced column-panel layout forced on whatever comes (tiny tiles, several rounds on pretended 1-4 CUs, rectangular operators)"""
    rng = np.random.default_rng(seed)
    rows = int(rng.choice([1, 17, 70, 1000, 4097, 9001, 15000]))
    cols = rows if rng.random() < 0.6 else int(rows + rng.integers(1, 3_000_000))
    kind = str(rng.choice(["uniform", "band", "cluster", "mix"]))
    max_len = int(rng.integers(1, 40))
    dup = bool(rng.random() < 0.5)
    os.environ["SL_PW_FORCE"] = "1"
    cus = int(rng.integers(1, 5))
    os.environ["SL_PW_CUS"] = str(cus)
    xcd = cus in (2, 4) and rng.random() < 0.5                    # spans dealt inside one L2 group (+ edge-first order where an interior is left)
    if xcd:
        os.environ["SL_PW_XCD"] = "2"
    try:
        rp, ci, va = F._structured_system(rng, rows, cols, kind, max_len, dup)
        m = S.SparseMatrix.from_csr(rp, ci, va, rows, cols, column_panels=True)
        seen("paced", m)                 # (a handful of rows against millions of columns falls back to other layouts: recorded, not an error)
        for order in (0, 1):
            x = rng.standard_normal(cols)
            assert bits_equal(m.multiply_vector(x, order), O.spmv(rp, ci, va, x, order)), ("spmv", order)
        if rows == cols:
            b = rng.standard_normal(rows)
            g = S.NeumannSolver().solve(m, b, S.SolverOptions(tolerance=1e-11))
            o = O.neumann_solve(rp, ci, va, b, tolerance=1e-11)
            assert (g.iterations, bool(g.converged)) == (o["iterations"], bool(o["converged"])) and bits_equal(g.solution, o["x"]), "neumann"
    finally:
        del os.environ["SL_PW_FORCE"], os.environ["SL_PW_CUS"]
        os.environ.pop("SL_PW_XCD", None)


def case_orderany(seed):
    """SL_ORDER_ANY on the forced order-free stream: within the reordering bound per row, the exact orders on the same matrix bit-exact"""
    rng = np.random.default_rng(seed)
    rows = int(rng.choice([64, 1000, 4097, 9001, 20000]))
    kind = str(rng.choice(["uniform", "cluster", "mix"]))
    max_len = int(rng.integers(1, 40))
    os.environ["SL_PW_FORCE"] = "1"
    os.environ["SL_PW_CUS"] = str(int(rng.integers(1, 5)))
    os.environ["SL_PWR_ROWS"] = str(int(rng.choice([64, 1024, 4096, 19968])))
    try:
        rp, ci, va = F._structured_system(rng, rows, rows, kind, max_len, bool(rng.random() < 0.5))
        m = S.SparseMatrix.from_csr(rp, ci, va, rows, rows, column_panels=True, order_any=True)
        assert m.info().column_panels == 4, "order-free stream not built"
        seen("orderany", m)
        x = rng.standard_normal(rows)
        ref = O.spmv(rp, ci, va, x)
        y = m.multiply_vector(x, order=L.SL_ORDER_ANY)
        lens = np.diff(rp.astype(np.int64))
        bound = 2.0 * np.maximum(lens, 1) * EPS * O.spmv(rp, ci, np.abs(va), np.abs(x))
        assert (np.abs(y - ref) <= bound).all(), "relaxed spmv beyond the reordering bound"
        for order in (0, 1):
            assert bits_equal(m.multiply_vector(x, order), O.spmv(rp, ci, va, x, order)), ("exact order on an order-any matrix", order)
    finally:
        del os.environ["SL_PW_FORCE"], os.environ["SL_PW_CUS"], os.environ["SL_PWR_ROWS"]


def _random_graph(rng, n, mean_deg):
    deg = rng.integers(0, 2 * mean_deg + 1, size=n)
    deg[rng.random(n) < 0.05] = 0                                  # dangling nodes
    src = np.repeat(np.arange(n), deg)
    dst = rng.integers(0, n, size=src.size)
    wgt = rng.uniform(0.1, 2.0, size=src.size)
    return src, dst, wgt


def case_acl(seed):
    """order-exact ACL push (WorkQueue order) forward and backward, and with a target: the pushed sequence, the counts, every bit"""
    from sublinear_time_solver_amd import push_graph as PG
    rng = np.random.default_rng(seed)
    n = int(rng.choice([5, 50, 400, 3000]))
    src, dst, wgt = _random_graph(rng, n, int(rng.integers(1, 9)))
    g = PG.PushGraph.from_edges(n, zip(src.tolist(), dst.tolist(), wgt.tolist()))
    cfg = PG.ForwardPushConfig(alpha=float(rng.choice([0.15, 0.2, 0.5])), epsilon=float(rng.choice([1e-4, 1e-6])), max_pushes=100_000,
                               queue_threshold=float(rng.choice([1e-8, 1e-15])), adaptive_threshold=bool(rng.random() < 0.5))
    kw = dict(alpha=cfg.alpha, epsilon=cfg.epsilon, max_pushes=cfg.max_pushes, queue_threshold=cfg.queue_threshold, adaptive_threshold=cfg.adaptive_threshold)
    seeds = sorted(set(int(v) for v in rng.integers(0, n, size=int(rng.integers(1, 4)))))

    def same(r, o, what):
        assert (r.push_count, r.nodes_visited) == (o["push_count"], o["nodes_visited"]), (what, "counts", r.push_count, o["push_count"])
        assert (r.push_log == o["push_log"]).all(), (what, "push sequence")
        assert bits_equal(r.estimate, o["estimate"]) and bits_equal(r.residual, o["residual"]), (what, "vectors")

    same(g.acl_forward(seeds, cfg, log_cap=100_000), O.acl_push(g.row_ptr, g.col_idx, g.weights, seeds, log_cap=100_000, **kw), "forward")
    same(g.acl_backward(seeds[:1], cfg, log_cap=100_000), O.acl_push(g.row_ptr, g.col_idx, g.weights, seeds[:1], backward=True, log_cap=100_000, **kw), "backward")
    target = int(rng.integers(0, n))
    same(g.acl_forward_with_target(seeds[0], target, 1e-3, cfg, log_cap=100_000),
         O.acl_push(g.row_ptr, g.col_idx, g.weights, seeds[:1], target=target, target_precision=1e-3, log_cap=100_000, **kw), "with target")


def case_southwell(seed):
    """TS solveForwardPush order (largest residual first): iteration count and the vectors"""
    rng = np.random.default_rng(seed)
    n = int(rng.choice([3, 40, 300, 2000]))
    rp, ci, va = F._random_system(rng, n, 0, int(rng.integers(2, 12)), False)
    b = rng.standard_normal(n) * (rng.random(n) < 0.2)
    if not b.any():
        b[0] = 1.0
    m = S.SparseMatrix.from_csr(rp, ci, va, n, n, with_transpose=True)
    eps = float(rng.choice([1e-4, 1e-8]))
    cap = int(rng.choice([50, 100_000]))
    o = O.ts_forward_push(rp, ci, va, b, eps, cap)
    g = S.GaussSouthwellSolver(epsilon=eps, max_iterations=cap).solve(m, b, on_failure="return")
    assert g["iterations"] == o["iterations"] and g["converged"] == o["converged"], ("southwell counts", g["iterations"], o["iterations"])
    assert bits_equal(g["solution"], o["x"]) and bits_equal(g["residual_vector"], o["r"]), "southwell vectors"


def case_cg(seed):
    """f-1 CG on random symmetric, strictly dominant systems: the oracle's iteration count, the solution to 1e-10"""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    n = int(rng.choice([10, 300, 3000]))
    rp, ci, va = F._random_system(rng, n, 0, int(rng.integers(2, 12)), False)
    a = sp.csr_matrix((va, ci.astype(np.int64), rp.astype(np.int64)), shape=(n, n))
    off = a - sp.diags(a.diagonal())
    sym = ((off + off.T) * 0.5).tocsr()
    sym = (sym + sp.diags(np.asarray(np.abs(sym).sum(axis=1)).ravel() * 1.5 + 1.0)).tocsr()
    sym.sort_indices()
    rp2, ci2, va2 = sym.indptr.astype(np.uint32), sym.indices.astype(np.uint32), sym.data.astype(np.float64)
    b = rng.standard_normal(n)
    m = S.SparseMatrix.from_csr(rp2, ci2, va2, n, n)
    g = S.ConjugateGradientSolver(tolerance=1e-9, max_iterations=500).solve(m, b)
    o = O.cg_solve(rp2, ci2, va2, b, tolerance=1e-9, max_iterations=500)
    assert abs(g.iterations - o["iterations"]) <= 1, ("cg iterations", g.iterations, o["iterations"])
    assert np.max(np.abs(g.solution - o["x"])) <= 1e-10 * max(1.0, np.max(np.abs(o["x"]))), "cg solution"


def case_wideband(seed):
    """bands wider than the LDS window (w >= 9500) — run this kind with SL_PW_BAND=<log2 panel width> in the environment to force the
    wide-band paced layout (the knob is read once per process); without it the library picks by size.  SpMV in both orders, one solve."""
    from sublinear_time_solver_amd import generators as G
    rng = np.random.default_rng(seed)
    n = int(rng.choice([30_000, 66_000, 150_001, 400_000]))
    w = int(rng.integers(9_500, min(n // 2 - 1, 70_000)))
    k = int(rng.choice([3, 5, 8, 16, 24]))
    lo = int(rng.integers(0, n // 3)) if rng.random() < 0.3 else 0
    hi = n if lo == 0 else int(rng.integers(lo + n // 3, n))
    rp, ci, va, b = G.sdd_rows(n, k, seed & 0xFFFF, w, lo, hi)
    forced = int(os.environ.get("SL_PW_BAND", "0") or 0) > 0
    if forced:                                                     # small systems: the layout has to be forced, on a pretended 1-4 CU device
        os.environ["SL_PW_FORCE"] = "1"
        os.environ["SL_PW_CUS"] = str(int(rng.integers(1, 5)))
    try:
        m = S.SparseMatrix.from_csr(rp, ci, va, hi - lo, n, row_offset=lo)
    finally:
        if forced:
            del os.environ["SL_PW_FORCE"], os.environ["SL_PW_CUS"]
    seen("wideband" + (" (row slice)" if (lo, hi) != (0, n) else ""), m)
    x = rng.standard_normal(n)
    for order in (0, 1):
        assert bits_equal(m.multiply_vector(x, order), O.spmv(rp, ci, va, x, order)), ("spmv", order, m.info().column_panels)
    if lo == 0 and hi == n:
        g = S.NeumannSolver().solve(m, b, S.SolverOptions(tolerance=1e-10))
        o = O.neumann_solve(rp, ci, va, b, tolerance=1e-10)
        assert (g.iterations, bool(g.converged)) == (o["iterations"], bool(o["converged"])) and bits_equal(g.solution, o["x"]), "neumann"


def case_session(seed):
    """query sessions: single estimates against the oracle's push on the transpose, batches on lanes against the single estimates"""
    rng = np.random.default_rng(seed)
    n = int(rng.choice([50, 700, 5000]))
    rp, ci, va = F._random_system(rng, n, 0, int(rng.integers(2, 14)), bool(rng.random() < 0.3) and n >= 700)
    b = rng.standard_normal(n)
    m = S.SparseMatrix.from_csr(rp, ci, va, n, n, with_transpose=True)
    trp, tci, tva = O.csr_transpose(rp, ci, va, n)
    theta = float(rng.choice([1e-4, 1e-8]))
    rows = [int(v) for v in rng.integers(0, n, size=int(rng.integers(1, 12)))]
    with S.QuerySession(m, b) as sess:
        one = [sess.estimate(r, theta=theta) for r in rows]
        for r, e in zip(rows[:3], one[:3]):
            unit = np.zeros(n)
            unit[r] = 1.0
            q = O.push_sync_solve(trp, tci, tva, unit, theta=theta)
            assert (e.rounds, e.pushes) == (q["rounds"], q["pushes"]), ("query counts", r)
            assert abs(e.estimate - float(np.dot(q["x"], b))) <= 1e-13 * max(1.0, np.abs(q["x"]).sum() * np.abs(b).max()), ("query value", r)
        got = sess.estimate_batch(rows, theta=theta, lanes=int(rng.integers(1, 9)))
        for a, g in zip(one, got):
            assert np.float64(a.estimate).view(np.uint64) == np.float64(g.estimate).view(np.uint64), "batch value"
            assert (a.rounds, a.pushes, a.rows_touched, a.converged) == (g.rounds, g.pushes, g.rows_touched, g.converged), "batch counts"


def case_trait(seed):
    """the element / iterator / norm side of trait Matrix on random structures with duplicated columns and hub rows: get (incl. misses and
    out of bounds), row_iter, col_iter, frobenius_norm, sparsity_info, spmv_add — exact except the norm"""
    rng = np.random.default_rng(seed)
    n = int(rng.choice([1, 2, 63, 64, 65, 130, 900, 2500]))
    long_rows = bool(rng.random() < 0.5) and n >= 700
    rp, ci, va = F._random_system(rng, n, 0, int(rng.integers(1, 40)), long_rows)
    if rng.random() < 0.6 and n > 2:                           # duplicate some entries (from_triplets keeps them, sparse.rs:80-132)
        tr = np.repeat(np.arange(n), np.diff(rp.astype(np.int64)))
        pick = rng.random(tr.size) < 0.15
        reps = rng.integers(1, 4, size=int(pick.sum()))
        tr2 = np.concatenate([tr, np.repeat(tr[pick], reps)]); tc2 = np.concatenate([ci, np.repeat(ci[pick], reps)])
        tv2 = np.concatenate([va, rng.standard_normal(int(reps.sum()))])
        rp, ci, va = O.csr_from_triplets(tr2, tc2, tv2, n, n)
    m = S.SparseMatrix.from_csr(rp, ci, va, n, n)
    seen("trait", m)
    for r in set(int(v) for v in rng.integers(0, n, size=6)) | {0, n - 1, int(np.argmax(np.diff(rp.astype(np.int64))))}:
        co, vo = O.csr_row(rp, ci, va, r)
        got = list(m.row_iter(r))
        assert [c for c, _ in got] == co.tolist() and bits_equal([v for _, v in got], vo), ("row_iter", r)
        for c in list(dict.fromkeys(co.tolist()))[:12] + [int(v) for v in rng.integers(0, n, size=2)] + [n]:
            want, have = O.matrix_get(rp, ci, va, r, c), m.get(r, c)
            assert (want is None) == (have is None) and (want is None or bits_equal([want], [have])), ("get", r, c)
    for c in set(int(v) for v in rng.integers(0, n, size=3)) | {int(np.bincount(ci, minlength=n).argmax()) if ci.size else 0}:
        ro, vo = O.csr_col(rp, ci, va, c)
        got = list(m.col_iter(c))
        assert [r for r, _ in got] == ro.tolist() and bits_equal([v for _, v in got], vo), ("col_iter", c)
    want = O.frobenius_norm(rp, va)
    assert abs(m.frobenius_norm() - want) <= 1e-12 * max(want, 1e-300), "frobenius"
    assert m.sparsity_info() == O.sparsity_info(rp, ci), "sparsity_info"
    x, y0 = rng.standard_normal(n), rng.standard_normal(n)
    y = y0.copy()
    m.multiply_vector_add(x, y)
    assert bits_equal(y, O.spmv_add(rp, ci, va, x, y0)), "spmv_add"


def case_walk(seed):
    """solveRandomWalk and the single-entry walks: per-coordinate means / variances against the CPU checker's block form, per-walk values bit for bit"""
    rng = np.random.default_rng(seed)
    n = int(rng.choice([3, 40, 150]))
    rp, ci, va = F._random_system(rng, n, 0, int(rng.integers(2, 9)), False)
    b = rng.standard_normal(n)
    m = S.SparseMatrix.from_csr(rp, ci, va, n, n, keep_csr=True)
    W, sd = int(rng.choice([100, 257, 1000])), int(rng.integers(0, 2 ** 32))
    r = S.random_walk_solve(m, b, 0.1, sd, num_walks=W)
    o = O.ts_random_walk_solve(rp, ci, va, b, 0.1, sd, num_walks=W, per_walk_streams=True)
    scale = max(np.abs(o["x"]).max(), 1e-300)
    assert np.abs(r["solution"] - o["x"]).max() <= 1e-12 * scale and r["converged"] == o["converged"] and r["num_walks"] == W, "walk solve"
    assert np.abs(r["variances"] - o["variances"]).max() <= 1e-11 * max(o["variances"].max(), scale * scale), "walk variances"
    row = int(rng.integers(0, n))
    import ctypes as C
    vals, res = np.zeros(W), L.WalkResult()
    L.check(L.load().sl_estimate_entry_random_walk(m._h, L.ptr(b), 0, row, 0.1, sd, 0, W, L.ptr(vals), C.byref(res)))
    assert bits_equal(vals, O.ts_random_walk_streams(rp, ci, va, b, row, W, sd)[0]), "walk values"


def case_mutate(seed):
    """the in-place mutators on random structures (duplicated columns, hub rows, rows without a diagonal, with and without raw CSR /
    transpose, forced paced panels now and then): a random sequence of scale / add_diagonal, then every layout copy against the CPU
    checker applied to the plain CSR — SpMV in both orders, row_iter, the transpose, a full solve where the result is still dominant"""
    rng = np.random.default_rng(seed)
    n = int(rng.choice([1, 2, 63, 65, 130, 900, 2500, 6000]))
    long_rows = bool(rng.random() < 0.4) and n >= 700
    rp, ci, va = F._random_system(rng, n, int(rng.choice([0, 0, 40])) if n > 100 else 0, int(rng.integers(1, 30)), long_rows)
    tr = np.repeat(np.arange(n), np.diff(rp.astype(np.int64)))
    keep = np.ones(tr.size, dtype=bool)
    if rng.random() < 0.5 and n > 4:                               # some rows lose their diagonal entry: add_diagonal must skip them
        keep &= ~((tr == ci) & (rng.random(tr.size) < 0.2))
    tr2, tc2, tv2 = tr[keep], ci[keep], va[keep]
    if rng.random() < 0.5 and n > 2:                               # some diagonals stored twice: ONE of them changes
        d = np.flatnonzero(tr2 == tc2)
        d = d[rng.random(d.size) < 0.2]
        tr2, tc2, tv2 = np.concatenate([tr2, tr2[d]]), np.concatenate([tc2, tc2[d]]), np.concatenate([tv2, rng.standard_normal(d.size)])
    rp, ci, va = O.csr_from_triplets(tr2, tc2, tv2, n, n)
    mode = int(rng.integers(0, 3))
    forced = n >= 2000 and rng.random() < 0.4
    if forced:
        os.environ["SL_PW_FORCE"], os.environ["SL_PW_CUS"] = "1", "2"
    try:
        m = S.SparseMatrix.from_csr(rp, ci, va, n, n, with_transpose=mode == 2, keep_csr=mode >= 1, column_panels=True if forced else None)
    finally:
        os.environ.pop("SL_PW_FORCE", None); os.environ.pop("SL_PW_CUS", None)
    seen("mutate", m)
    want = va.copy()
    for _ in range(int(rng.integers(1, 5))):
        if rng.random() < 0.5:
            f = float(rng.choice([2.0, -0.5, 1.0 / 3.0, 1e-3, 7.25]))
            os.environ.update({"SL_PW_FORCE": "1", "SL_PW_CUS": "2"}) if forced else None
            try:
                m.scale(f)
            finally:
                os.environ.pop("SL_PW_FORCE", None); os.environ.pop("SL_PW_CUS", None)
            want = O.csr_scale(want, f)
        else:
            a = float(rng.choice([0.5, -0.125, 3.0, 1e-9]))
            os.environ.update({"SL_PW_FORCE": "1", "SL_PW_CUS": "2"}) if forced else None
            try:
                m.add_diagonal(a)
            finally:
                os.environ.pop("SL_PW_FORCE", None); os.environ.pop("SL_PW_CUS", None)
            want, _ = O.csr_add_diagonal(rp, ci, want, a)
    x = rng.standard_normal(n)
    for order in (0, 1):
        assert bits_equal(m.multiply_vector(x, order), O.spmv(rp, ci, want, x, order)), ("spmv after mutation", order)
    for r in set(int(v) for v in rng.integers(0, n, size=4)) | {0, n - 1}:
        got = list(m.row_iter(r))
        co, vo = O.csr_row(rp, ci, want, r)
        assert [c for c, _ in got] == co.tolist() and bits_equal([v for _, v in got], vo), ("row_iter after mutation", r)
    if mode == 2 and n > 1:
        mt = m.transpose(keep_csr=True)
        _, tci, tva = O.csr_transpose(rp, ci, want, n)
        y = rng.standard_normal(n)
        trp = O.csr_transpose(rp, ci, want, n)[0]
        assert bits_equal(mt.multiply_vector(y), O.spmv(trp, tci, tva, y)), "transpose after mutation"


def case_walkserial(seed):
    """SL_WALK_STREAM_SERIAL against the CPU checker's serial form (= the reference as written): per-walk values, mean, variance, and the
    solve's x / variances / total variance — bit for bit"""
    import ctypes as C
    rng = np.random.default_rng(seed)
    n = int(rng.choice([3, 40, 150]))
    rp, ci, va = F._random_system(rng, n, 0, int(rng.integers(2, 9)), False)
    b = rng.standard_normal(n)
    m = S.SparseMatrix.from_csr(rp, ci, va, n, n, keep_csr=True)
    W, sd, row = int(rng.choice([100, 257, 700])), int(rng.integers(0, 2 ** 32)), int(rng.integers(0, n))
    vals, res = np.zeros(W), L.WalkResult()
    L.check(L.load().sl_estimate_entry_random_walk(m._h, L.ptr(b), 0, row, 0.1, sd, L.SL_WALK_STREAM_SERIAL, W, L.ptr(vals), C.byref(res)))
    ov, om, ovar = O.ts_random_walk_serial(rp, ci, va, b, row, W, sd)
    assert bits_equal(vals, ov) and (res.estimate, res.variance) == (om, ovar), "serial walk values"
    r = S.random_walk_solve(m, b, 0.1, sd, num_walks=W, stream="reference")
    o = O.ts_random_walk_solve(rp, ci, va, b, 0.1, sd, num_walks=W, per_walk_streams=False)
    assert bits_equal(r["solution"], o["x"]) and bits_equal(r["variances"], o["variances"]) and r["total_variance"] == o["total_variance"], "serial walk solve"


KINDS = {"mutate": case_mutate, "walkserial": case_walkserial, "trait": case_trait, "walk": case_walk, "general": case_general, "paced": case_paced, "orderany": case_orderany, "acl": case_acl, "southwell": case_southwell, "cg": case_cg,
         "wideband": case_wideband, "session": case_session}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--kinds", default="general,paced,orderany,acl,southwell,cg")
    ap.add_argument("--seed0", type=int, default=int(time.time()) & 0xFFFFFF)
    ap.add_argument("--replay", default=None, help='"<kind> <seed>"')
    args = ap.parse_args()
    if args.replay:
        kind, seed = args.replay.split()
        KINDS[kind](int(seed))
        print("replay ok")
        return 0
    kinds = [k for k in args.kinds.split(",") if k]
    done = {k: 0 for k in kinds}
    failures = []
    t_end = time.time() + args.seconds
    seed = args.seed0
    while time.time() < t_end and len(failures) < 20:
        for k in kinds:
            seed += 1
            try:
                KINDS[k](seed)
            except Exception as e:                                 # noqa: BLE001 — a campaign reports and goes on
                failures.append({"kind": k, "seed": seed, "error": f"{type(e).__name__}: {e}"[:300]})
                print(f"FAIL {k} {seed}: {type(e).__name__}: {e}", file=sys.stderr)
                traceback.print_exc(limit=3, file=sys.stderr)
            done[k] += 1
            if time.time() >= t_end:
                break
    print(json.dumps({"seed0": args.seed0, "cases": done, "failures": failures, "layouts_seen": dict(sorted(SEEN.items()))}))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
