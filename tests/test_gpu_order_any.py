"""SL_ORDER_ANY on the order-free column stream (SL_MATRIX_ORDER_ANY; sl_pwr_kernel): the opt-in mode in which row sums are added
in whatever order the device finds fastest.  Parity gate = BASELINE.json north_star / BASELINE.md §3: max|x_gpu - x_cpu| / ||x||_inf
<= 1e-10 and the oracle's iteration counts; per launch the row results must equal the oracle's to a few ulps of sum|a_ij x_j|
(a reordered sum of k terms differs from the sequential one by at most (k - 1) eps sum|terms|).  The exact orders on the same matrix
stay bit-exact (they take the row-slice kernels), and the push never runs the relaxed path."""
import ctypes as C
import os

import numpy as np
import pytest

import sublinear_time_solver_amd as S
from sublinear_time_solver_amd import _lib as L
from sublinear_time_solver_amd import generators as G
from oracle import oracle as O

pytestmark = pytest.mark.gpu
EPS = np.finfo(np.float64).eps


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.fixture
def forced_small_device(monkeypatch):
    """build the order-free stream on small systems, on a pretended 3-CU device: several rounds of block tiles"""
    monkeypatch.setenv("SL_PW_FORCE", "1")
    monkeypatch.setenv("SL_PW_CUS", "3")
    monkeypatch.setenv("SL_PWR_ROWS", "4096")
    yield


def abs_spmv(rp, ci, va, x):
    return O.spmv(rp, ci, np.abs(va), np.abs(x))


def ragged_system(n, seed, hubs=True, dups=True):
    """rows of 1..40 entries, a few hub rows of thousands, duplicate (row, col) entries, strictly row dominant"""
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, 40, size=n)
    if hubs:
        lens[rng.integers(0, n, size=5)] = rng.integers(2000, 6000, size=5)
    rows = np.repeat(np.arange(n), lens)
    cols = rng.integers(0, n, size=rows.size)
    vals = rng.uniform(-1, 1, size=rows.size)
    if dups:
        d = rng.integers(0, rows.size, size=rows.size // 50)
        rows, cols, vals = np.concatenate([rows, rows[d]]), np.concatenate([cols, cols[d]]), np.concatenate([vals, rng.uniform(-1, 1, size=d.size)])
    keep = rows != cols
    rows, cols, vals = rows[keep], cols[keep], vals[keep]
    off = np.zeros(n)
    np.add.at(off, rows, np.abs(vals))
    rows = np.concatenate([rows, np.arange(n)]); cols = np.concatenate([cols, np.arange(n)]); vals = np.concatenate([vals, 2.0 * off + 1.0])
    order = np.lexsort((cols, rows))
    rows, cols, vals = rows[order], cols[order], vals[order]
    rp = np.zeros(n + 1, dtype=np.uint32)
    np.add.at(rp, rows + 1, 1)
    rp = np.cumsum(rp).astype(np.uint32)
    b = 1.0 + 0.001 * (np.arange(n) % 1000)
    return rp, cols.astype(np.uint32), vals, b


@pytest.mark.parametrize("n,k,seed", [(50_003, 16, 3), (20_000, 8, 5), (131_072, 16, 9)])
def test_order_any_on_uniform_columns(gpu, forced_small_device, n, k, seed):
    rp, ci, va, b = G.sdd_rows(n, k, seed=seed)
    m = S.SparseMatrix.from_csr(rp, ci, va, n, n, column_panels=True, order_any=True)
    assert m.info().column_panels == 4, "the order-free column stream was not built"
    x = np.cos(np.arange(n) * 0.37) * (1.0 + (np.arange(n) % 7))
    y = m.multiply_vector(x, order=L.SL_ORDER_ANY)
    ref = O.spmv(rp, ci, va, x)
    assert (np.abs(y - ref) <= 4 * k * EPS * abs_spmv(rp, ci, va, x)).all()
    assert (bits(m.multiply_vector(x)) == bits(ref)).all(), "the CSR order on an order-any matrix must stay bit-exact"
    o = O.neumann_solve(rp, ci, va, b, tolerance=1e-10)
    g = S.NeumannSolver(order=L.SL_ORDER_ANY).solve(m, b, S.SolverOptions(tolerance=1e-10))
    assert g.converged and g.iterations == o["iterations"]
    assert np.max(np.abs(g.solution - o["x"])) <= 1e-10 * np.max(np.abs(o["x"]))
    assert abs(g.residual_norm - o["residual_norm"]) <= 1e-6 * o["residual_norm"] + 1e-14
    e = S.NeumannSolver().solve(m, b, S.SolverOptions(tolerance=1e-10))
    assert e.iterations == o["iterations"] and (bits(e.solution) == bits(o["x"])).all()


def test_order_any_on_ragged_rows_with_hubs_and_duplicates(gpu, forced_small_device):
    n = 60_000
    rp, ci, va, b = ragged_system(n, 11)
    m = S.SparseMatrix.from_csr(rp, ci, va, n, n, column_panels=True, order_any=True)
    assert m.info().column_panels == 4
    x = np.sin(np.arange(n) * 0.11) + 2.0
    y = m.multiply_vector(x, order=L.SL_ORDER_ANY)
    ref = O.spmv(rp, ci, va, x)
    maxlen = int(np.max(np.diff(rp.astype(np.int64))))
    assert (np.abs(y - ref) <= 2 * maxlen * EPS * abs_spmv(rp, ci, va, x)).all()
    o = O.neumann_solve(rp, ci, va, b, tolerance=1e-10)
    g = S.NeumannSolver(order=L.SL_ORDER_ANY).solve(m, b, S.SolverOptions(tolerance=1e-10))
    assert g.converged and g.iterations == o["iterations"]
    assert np.max(np.abs(g.solution - o["x"])) <= 1e-10 * np.max(np.abs(o["x"]))


def test_order_any_on_a_row_slice(gpu, forced_small_device):
    """one rank's rows of a larger system: row_offset != 0, n_cols > n_rows — the diagonal sits at column row_offset + i"""
    n_global, k, lo, hi = 200_000, 16, 70_000, 130_000
    rp, ci, va, b = G.sdd_rows(n_global, k, 4, 0, lo, hi)
    m = S.SparseMatrix.from_csr(rp, ci, va, hi - lo, n_global, row_offset=lo, column_panels=True, order_any=True)
    assert m.info().column_panels == 4
    x = np.cos(np.arange(n_global) * 0.013) + 1.5
    y = m.multiply_vector(x, order=L.SL_ORDER_ANY)
    ref = O.spmv(rp, ci, va, x)
    assert (np.abs(y - ref) <= 4 * k * EPS * abs_spmv(rp, ci, va, x)).all()


def test_push_on_an_order_any_matrix_stays_bit_exact(gpu, forced_small_device):
    n = 40_000
    rp, ci, va, b = G.sdd_rows(n, 16, seed=21)
    m = S.SparseMatrix.from_csr(rp, ci, va, n, n, with_transpose=True, column_panels=True, order_any=True)
    bs = b * (np.arange(n) % 5 == 0)
    p = S.PushSolver(theta=1e-8, order=L.SL_ORDER_ANY).solve(m, bs, log_frontier=1 << 20)
    q = O.push_sync_solve(rp, ci, va, bs, theta=1e-8, log_cap=1 << 20)
    assert p["rounds"] == q["rounds"] and (p["frontier_log"] == q["frontier_log"]).all() and (bits(p["solution"]) == bits(q["x"])).all()


def test_order_any_without_the_stream_runs_the_csr_order(gpu):
    n = 30_000
    rp, ci, va, b = G.sdd_rows(n, 16, seed=2, half_bandwidth=500)
    m = S.SparseMatrix.from_csr(rp, ci, va, n, n, order_any=True)      # a band: column panels do not pay, nothing order-free is built
    assert m.info().column_panels == 0
    x = np.cos(np.arange(n) * 0.37)
    assert (bits(m.multiply_vector(x, order=L.SL_ORDER_ANY)) == bits(O.spmv(rp, ci, va, x))).all()


def test_order_any_headline_instance_sampled(gpu):
    """bench.py's headline instance in the relaxed mode at its own size: n = 10^7, 16 per row, uniform columns — sampled row blocks of
    one fused step against the oracle within the reordering bound, the norm against the host sum"""
    import torch
    lib = L.load()
    dev = torch.device("cuda", 0)
    n, k, seed = 10_000_000, 16, 1
    rp = torch.empty(n + 1, dtype=torch.int32, device=dev)
    ci = torch.empty(n * k, dtype=torch.int32, device=dev)
    va = torch.empty(n * k, dtype=torch.float64, device=dev)
    b = torch.empty(n, dtype=torch.float64, device=dev)
    L.check(lib.sl_synth_sdd_device(n, k, seed, 0, 0, n, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), b.data_ptr()))
    h = C.c_void_p()
    L.check(lib.sl_matrix_create_csr(n, n, n * k, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), L.SL_MEM_DEVICE, 0, L.SL_MATRIX_ORDER_ANY, C.byref(h)))
    del rp, ci, va
    torch.cuda.empty_cache()
    try:
        info = L.MatrixInfo()
        L.check(lib.sl_matrix_get_info(h, C.byref(info)))
        assert info.column_panels == 4
        dinv = torch.empty(n, dtype=torch.float64, device=dev)
        L.check(lib.sl_matrix_diagonal_inverse(h, dinv.data_ptr(), L.SL_MEM_DEVICE))
        t0 = b * dinv
        x = t0.clone()
        t1 = torch.zeros(n, dtype=torch.float64, device=dev)
        norm2 = torch.zeros(2, dtype=torch.float64, device=dev)
        L.check(lib.sl_neumann_step(h, dinv.data_ptr(), t0.data_ptr(), t1.data_ptr(), x.data_ptr(), norm2.data_ptr(), L.SL_ORDER_ANY))
        L.check(lib.sl_synchronize())
        t0h, t1h, xh = t0.cpu().numpy(), t1.cpu().numpy(), x.cpu().numpy()
        for a in (0, 4_999_936, n - 4096, 1_234_560):
            e = a + 4096
            rrp, rci, rva, _ = G.sdd_rows(n, k, seed, 0, a, e)
            y = O.spmv(rrp, rci, rva, t0h)
            d = 10.0 + 0.01 * (np.arange(a, e) % 1000)
            tn = t0h[a:e] - y * (1.0 / d)
            bound = 4 * k * EPS * abs_spmv(rrp, rci, rva, t0h) / d + 4 * EPS * np.abs(t0h[a:e])
            assert (np.abs(t1h[a:e] - tn) <= bound).all(), f"rows {a}..{e}"
            assert (np.abs(xh[a:e] - (t0h[a:e] + tn)) <= bound + 2 * EPS * np.abs(xh[a:e])).all()
        assert abs(float(norm2[0]) - float(np.dot(t1h, t1h))) <= 1e-10 * float(norm2[0])
        ta, tb, xx = t0.clone(), torch.empty_like(t0), t0.clone()
        ms = C.c_float(0)
        L.check(lib.sl_neumann_run_steps(h, dinv.data_ptr(), ta.data_ptr(), tb.data_ptr(), xx.data_ptr(), norm2.data_ptr(), L.SL_ORDER_ANY, 3, C.byref(ms)))
        L.check(lib.sl_neumann_run_steps(h, dinv.data_ptr(), tb.data_ptr(), ta.data_ptr(), xx.data_ptr(), norm2.data_ptr(), L.SL_ORDER_ANY, 20, C.byref(ms)))
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        if os.path.isdir(out):
            with open(os.path.join(out, "fullsize_steps.jsonl"), "a") as f:
                f.write('{"instance": "n=1e7 k=16 uniform, SL_ORDER_ANY (order-free column stream)", "ms_per_step": %.4f}\n' % (ms.value / 20))
    finally:
        lib.sl_matrix_destroy(h)
