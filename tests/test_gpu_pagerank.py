"""GPU tests for the PageRank / estimateEntry side of the path (BASELINE config 4 at test size) and for the
uniform-width band-kernel variants (16-bit column offsets, pipelining)."""
import ctypes as C

import numpy as np
import pytest

import sublinear_time_solver_amd as S
from sublinear_time_solver_amd import _lib as L
from sublinear_time_solver_amd import generators as G
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _bits_equal(a, b):
    return (np.ascontiguousarray(a).view(np.uint64) == np.ascontiguousarray(b).view(np.uint64)).all()


def _spr(n, seed=1, alpha=0.85):
    import torch
    lib = L.load()
    rp = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    nnz = C.c_uint64(0)
    L.check(lib.sl_synth_pagerank_device(n, seed, alpha, 2, 8192, rp.data_ptr(), None, None, C.byref(nnz)))
    ci = torch.empty(nnz.value, dtype=torch.int32, device="cuda")
    va = torch.empty(nnz.value, dtype=torch.float64, device="cuda")
    L.check(lib.sl_synth_pagerank_device(n, seed, alpha, 2, 8192, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), C.byref(nnz)))
    L.check(lib.sl_synchronize())
    return rp, ci, va


def test_spr_generator_and_transposed_query(gpu):
    n, alpha = 50_000, 0.85
    rp, ci, va = _spr(n)
    M = S.SparseMatrix.from_csr(rp, ci, va, n, n, with_transpose=True, keep_csr=True, device=True)
    hrp, hci, hva = M.to_csr()
    deg = np.diff(hrp.astype(np.int64)) - 1
    assert deg.min() >= 2 and deg.max() <= 8192 and 8 < deg.mean() < 30           # power-law out-degrees, mean ~16
    assert (hci[hrp[:-1]] == np.arange(n)).all() and (hva[hrp[:-1]] == 1.0).all()  # diagonal first, = 1
    rows = np.repeat(np.arange(n), deg + 1)
    offd = hci != rows
    assert np.allclose(np.add.reduceat(np.where(offd, -hva, 0.0), hrp[:-1]), alpha)  # rows of P sum to 1
    assert M.is_diagonally_dominant()                                               # M = I - alpha P is ROW dominant
    # A = M^T is the PageRank solve matrix (column dominant): the transpose object equals the oracle's transpose
    A = M.transpose(with_transpose=True, keep_csr=True)
    arp, aci, ava = A.to_csr()
    trp, tci, tva = O.csr_transpose(hrp, hci, hva, n)
    assert (arp == trp).all() and (aci == tci).all() and _bits_equal(ava, tva)
    b = np.full(n, (1.0 - alpha) / n)
    # full solve of A x = b: GPU push vs oracle push, bit for bit (dense + sparse rounds, ragged power-law rows)
    g = S.PushSolver(theta=1e-11 / n).solve(A, b)
    o = O.push_sync_solve(arp, aci, ava, b, theta=1e-11 / n)
    assert g["converged"] and g["rounds"] == o["rounds"] and g["pushes"] == o["pushes"]
    assert _bits_equal(g["solution"], o["x"])
    x = g["solution"]
    assert abs(x.sum() - 1.0) < 1e-6                                                # PageRank mass
    # single-entry queries: push on A^T = M (given directly) and on A (transpose taken inside) agree with x
    for row in (0, 1, n // 2, n - 1):
        e1 = S.estimate_entry(M, b, row, theta=1e-12, matrix_is_transpose=True)
        e2 = S.estimate_entry(A, b, row, theta=1e-12)
        for e in (e1, e2):
            assert e.converged and abs(e.estimate - x[row]) <= e.residual_l1 * np.abs(x).max() + 1e-15
        assert e1.estimate == e2.estimate and e1.pushes == e2.pushes               # same arithmetic either way
    # reference semantics of the push family on this graph (forward_push.rs:67-216): ACL estimate from node 0
    # solves the same system with b = alpha' e_0 on the row-stochastic side; mass is conserved
    acl = O.acl_push(hrp, hci, np.where(offd, -hva / alpha, 0.0), [0], alpha=0.15, epsilon=1e-7)
    assert abs(acl["estimate"].sum() + acl["residual"].sum() - 1.0) < 1e-9


@pytest.mark.parametrize("n,k,w", [(40_000, 16, 500), (40_000, 16, 4096), (30_011, 8, 100), (30_011, 8, 3000),
                                   (30_011, 13, 700), (30_011, 5, 60), (30_011, 12, 2000), (9_000, 37, 900)])
def test_band_kernel_variants_bitwise(gpu, n, k, w):
    """band matrices, uniform width (unrolled path, octet 16-bit offsets) and other widths (batched ragged path, quad
    16-bit offsets), both summation orders, both epilogue families"""
    rp, ci, va, b = G.sdd_rows(n, k, seed=3, half_bandwidth=w)
    m = S.SparseMatrix.from_csr(rp, ci, va, n, n, with_transpose=True)
    info = m.info()
    assert info.uniform_width == (k if k % 4 == 0 else 0) and info.bandwidth <= w
    x = np.cos(np.arange(n))
    bs = b * (np.arange(n) % 3 == 0)
    for order in (0, 1):
        assert _bits_equal(m.multiply_vector(x, order), O.spmv(rp, ci, va, x, order))
        g = S.NeumannSolver(order=order).solve(m, b, S.SolverOptions(tolerance=1e-11))
        o = O.neumann_solve(rp, ci, va, b, tolerance=1e-11, order=order)
        assert g.iterations == o["iterations"] and _bits_equal(g.solution, o["x"])
        p = S.PushSolver(theta=1e-9, dense_switch=1e-9, order=order).solve(m, bs, log_frontier=4_000_000)      # dense rounds only
        q = O.push_sync_solve(rp, ci, va, bs, theta=1e-9, order=order, log_cap=4_000_000)
        assert p["rounds"] == q["rounds"] and (p["frontier_log"] == q["frontier_log"]).all()
        assert _bits_equal(p["solution"], q["x"]) and _bits_equal(p["residual"], q["r"])


def test_c4_full_size_pagerank_queries(gpu):
    """BASELINE config 4 at its full size (n = 10^7, ~1.1e8 edges): single-entry queries on a session against the full
    solve of the same system — size-independent properties: PageRank mass 1, every estimate inside its own error
    bound |x_row - est| <= ||r_y||_1 * ||x||_inf, repeated queries bit-identical, local queries touch a small part of
    the graph."""
    import torch
    n, alpha = 10_000_000, 0.85
    lib = L.load()
    rp, ci, va = _spr(n, seed=1, alpha=alpha)
    M = S.SparseMatrix.from_csr(rp, ci, va, n, n, with_transpose=True, device=True)
    del rp, ci, va
    b = torch.full((n,), (1.0 - alpha) / n, dtype=torch.float64, device="cuda")
    A = M.transpose(with_transpose=True)
    o = L.PushOptions()
    lib.sl_push_options_default(C.byref(o))
    o.theta, o.max_rounds, o.mem = 1e-13 / n, 10000, L.SL_MEM_DEVICE
    x = torch.zeros(n, dtype=torch.float64, device="cuda")
    res = L.PushResult()
    L.check(lib.sl_push_solve(A._h, b.data_ptr(), C.byref(o), x.data_ptr(), None, None, 0, None, C.byref(res)))
    assert res.converged and abs(float(x.sum()) - 1.0) < 1e-9
    xinf = float(x.abs().max())
    del A
    torch.cuda.empty_cache()
    with S.QuerySession(M, b, matrix_is_transpose=True, device=True) as q:
        for row in (n - 1, n // 2, 123_456, 0):
            for theta in (1e-5, 1e-7):
                e = q.estimate(row, theta=theta)
                xv = float(x[row])
                assert e.converged and abs(e.estimate - xv) <= e.residual_l1 * xinf + 1e-18, (row, theta)
                e2 = q.estimate(row, theta=theta)
                assert (e2.estimate, e2.residual_l1, e2.rounds, e2.pushes) == (e.estimate, e.residual_l1, e.rounds, e.pushes)
        local = q.estimate(n - 1, theta=1e-5)
        assert local.rows_touched < n // 50                       # a local query: < 2 % of the rows over all its rounds


def test_g9_gpu_pagerank_matches_the_reference_power_iteration(gpu):
    """the same goldens (reference Python power iteration, tests/golden/reference_pagerank.npz) through the GPU push"""
    import scipy.sparse as sp
    from pathlib import Path
    z = np.load(Path(__file__).resolve().parent / "golden" / "reference_pagerank.npz")
    for key in (str(c) for c in z["__cases"]):
        n, d = int(z[f"{key}__n"][0]), float(z[f"{key}__damping"][0])
        A = sp.csr_matrix((z[f"{key}__vals"], (z[f"{key}__rows"].astype(np.int64), z[f"{key}__cols"].astype(np.int64))), shape=(n, n))
        rp, ci, va, b = G.pagerank_system(n, A.indptr.astype(np.uint32), A.indices.astype(np.uint32), A.data, d)
        m = S.SparseMatrix.from_csr(rp, ci, va, n, n, with_transpose=True)
        g = S.PushSolver(theta=1e-18).solve(m, b)
        ref = z[f"{key}__pagerank"]
        assert g["converged"] and np.abs(g["solution"] - ref).max() <= 1e-13, key
        top = int(np.argmax(ref))
        e = S.estimate_entry(m, b, top, theta=1e-16)
        assert abs(e.estimate - ref[top]) <= 1e-12


def test_compute_pagerank_of_the_ts_surface(gpu):
    """SublinearSolver.compute_pagerank = computePageRank (core/solver.ts:664-722): adjacency in the TS wire format, the system assembled
    in CSR, solved by the solver's own method; against the same goldens, plus the reference's argument checks"""
    from pathlib import Path
    z = np.load(Path(__file__).resolve().parent / "golden" / "reference_pagerank.npz")
    key = str(z["__cases"][0])
    n, d = int(z[f"{key}__n"][0]), float(z[f"{key}__damping"][0])
    adj = {"rows": n, "cols": n, "format": "coo", "values": z[f"{key}__vals"].tolist(), "rowIndices": z[f"{key}__rows"].astype(int).tolist(),
           "colIndices": z[f"{key}__cols"].astype(int).tolist()}
    ref = z[f"{key}__pagerank"]
    for method in ("forward-push", "bidirectional"):          # (I - d P^T is COLUMN dominant: the Neumann path wants row dominance, neumann.rs:139-170)
        x = S.SublinearSolver(method=method, epsilon=1e-14, max_iterations=100000, push_order="synchronous").compute_pagerank(adj, damping=d, epsilon=1e-14, max_iterations=100000)
        assert np.abs(x - ref).max() <= 1e-11, (method, np.abs(x - ref).max())
    # personalised right-hand side: linear in it
    e0 = np.zeros(n); e0[0] = 1.0
    s = S.SublinearSolver(method="forward-push", push_order="synchronous")
    x0 = s.compute_pagerank(adj, damping=d, epsilon=1e-14, max_iterations=100000, personalized=e0)
    x2 = s.compute_pagerank(adj, damping=d, epsilon=1e-14, max_iterations=100000, personalized=2.0 * e0)
    assert np.abs(x2 - 2.0 * x0).max() <= 1e-12 and x0[0] > 0
    with pytest.raises(S.SolverError, match="damping must be between 0 and 1"):
        s.compute_pagerank(adj, damping=1.5)
    with pytest.raises(S.SolverError, match="Adjacency matrix must be square"):
        s.compute_pagerank({"rows": 2, "cols": 3, "format": "dense", "data": [[0, 1, 0], [1, 0, 0]]})


def test_index_only_stream_of_column_constant_operators_keeps_the_bits(gpu):
    """SL_PW_INDEX_ONLY=1: for I - (1 - alpha) P^T of an UNWEIGHTED graph every off-diagonal entry of column u is -(1 - alpha) / deg_u and
    the diagonal is exactly 1; the dense push rounds then run the paced kernel on the index words of its stream alone and gather the
    ready-made products colval_u * delta_u.  Same product per entry, same order: x, r, rounds, pushes bit for bit as with the full stream —
    on a forced paced layout with several rounds of tiles, hub rows (long-row kernel) and dangling columns; a WEIGHTED graph is not such an
    operator and must not be taken for one."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    prog = r"""
import json, numpy as np
import sublinear_time_solver_amd as S
from sublinear_time_solver_amd import generators as G
out = {}
n = 40_000
for tag, weighted in (("unit", False), ("weighted", True)):
    rp, ci, w = G.pagerank_graph(n, 11)
    if weighted:
        w = 0.5 + (np.arange(w.size) % 7) * 0.25
    arp, aci, ava, b = G.pagerank_system(n, rp, ci, w, damping=0.85)
    m = S.SparseMatrix.from_csr(arp, aci, ava, n, n, with_transpose=True)
    out[tag + " layout"] = int(m.info().column_panels)
    for theta in (1e-9, 1e-6):
        p = S.PushSolver(theta=theta, dense_switch=1.0 / 64.0).solve(m, b)
        out[f"{tag} {theta}"] = [int(p["rounds"]), int(p["pushes"]), int(p["dense_rounds"]), int(p["solution"].view(np.uint64).sum() % (1 << 61)),
                                 int(p["residual"].view(np.uint64).sum() % (1 << 61)), int(np.float64(p["solution"][17]).view(np.uint64))]
print("RESULT " + json.dumps(out))
"""
    base = dict(os.environ, SL_COLUMN_PANELS="1", SL_PW_FORCE="1", SL_PW_CUS="4", SL_LOG="1")
    res, logs = {}, {}
    for mode in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", prog], cwd=root, capture_output=True, text=True, timeout=900, env=dict(base, SL_PW_INDEX_ONLY=mode))
        assert r.returncode == 0, r.stderr[-3000:]
        res[mode] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
        logs[mode] = r.stderr
    assert res["0"] == res["1"], {k: (res["0"][k], res["1"][k]) for k in res["0"] if res["0"][k] != res["1"][k]}
    assert res["1"]["unit layout"] == 2 and res["1"]["unit 1e-09"][2] > 3                       # paced layout, several dense rounds
    assert logs["1"].count("column-constant operator") == 1 and "column-constant operator" not in logs["0"]   # the unit graph only, and only when asked


def test_g13_compute_pagerank_equals_the_references_own_typescript(gpu):
    """SublinearSolver({method: 'forward-push'}).computePageRank on the device against what the reference's own TypeScript returned for the
    same adjacency (tests/golden/reference_ts_pagerank.npz, make_golden_ts_pagerank.py): the solution bit for bit — assembly arithmetic,
    dangling nodes, self loops, weights, a personalised right-hand side, and the push in the reference's visiting order"""
    from tests.test_oracle_golden import golden_ts_pagerank_cases
    for k, g, c in golden_ts_pagerank_cases():
        n = c["n"]
        adj = {"rows": n, "cols": n, "format": "coo", "values": g[k + "/adj_values"].tolist(), "rowIndices": g[k + "/adj_rows"].tolist(), "colIndices": g[k + "/adj_cols"].tolist()}
        s = S.SublinearSolver(method="forward-push", epsilon=1e-6, max_iterations=10, push_order="reference")
        x = s.compute_pagerank(adj, damping=c["damping"], epsilon=c["eps"], max_iterations=c["maxit"], personalized=g[k + "/rhs"] if c["personalized"] else None)
        assert (np.ascontiguousarray(x).view(np.uint64) == g[k + "/solution"].view(np.uint64)).all(), k
