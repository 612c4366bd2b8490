"""N > 1 path on CPU: world_size-2 gloo run of the row-partitioned driver (sublinear_time_solver_amd.distributed).
The device kernel is replaced by a stand-in local step (the oracle's row loop on this rank's slice) — this
tests the HOST logic: partition bounds, all-gather / halo exchange, norm all-reduce, ping-pong buffers;
the partitioned iteration must reproduce the single-process one bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sublinear_time_solver_amd import distributed as D
from sublinear_time_solver_amd import generators as G
from oracle import oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_local_step(rp, ci, va, dinv, lo):
    """stand-in for sl_neumann_step on rows [lo, lo+rows): same arithmetic, via the oracle."""
    def step(t_in_full, t_out_local, x_local, norm2):
        t = t_in_full.numpy()
        y = O.spmv(rp, ci, va, t)
        tmp = y * dinv
        tn = t[lo:lo + tmp.size] - tmp
        t_out_local.copy_(torch.from_numpy(tn))
        x_local.add_(torch.from_numpy(tn))
        norm2[0] = float(np.dot(tn, tn))
    return step


def _worker(rank, world, port, n, k, w, steps, out_dir, split=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        part = D.RowPartition(n, world, rank)
        rp, ci, va, b = G.sdd_rows(n, k, 3, w, part.lo, part.hi)
        d = np.array([va[i * k:(i + 1) * k][ci[i * k:(i + 1) * k] == part.lo + i][0] for i in range(part.n_local)])
        dinv = 1.0 / d
        _, _, _, b_all = G.sdd_rows(n, k, 3, w)
        d_all = 10.0 + 0.01 * (np.arange(n) % 1000)
        t0 = np.zeros(part.n_padded)
        t0[:n] = b_all * (1.0 / d_all)
        t0 = torch.from_numpy(t0)
        x = t0[part.lo:part.hi].clone()
        ex = D.AllGatherExchange(part) if w == 0 else D.HaloExchange(part, w)
        if os.environ.get("SL_TEST_EXCHANGE") == "allreduce" and w:
            ex = D.HaloAllReduceExchange(part, w)
        local = _oracle_local_step(rp, ci, va, dinv, part.lo)
        if split:      # boundary rows first, exchange in flight under the interior rows
            def piece(lo_l, hi_l):
                prp = (rp[lo_l:hi_l + 1] - rp[lo_l]).astype(np.uint32)
                return (lo_l, hi_l, _oracle_local_step(prp, ci[rp[lo_l]:rp[hi_l]], va[rp[lo_l]:rp[hi_l]], dinv[lo_l:hi_l], part.lo + lo_l))
            bnd, inter = D.split_bounds(part.n_local, max(w, 64), rank > 0, rank < world - 1)
            local = D.SplitStep([piece(a, b) for a, b in bnd if b > a], [piece(a, b) for a, b in inter if b > a], torch.device("cpu"))
        every = int(os.environ.get("SL_TEST_REDUCE_EVERY", "1"))
        drv = D.PartitionedNeumann(part, local, ex, t0, x, reduce_every=every)
        if os.environ.get("SL_TEST_RESTART") == "1":      # bench.py's parity gate: a few steps, a look, back to the start
            t_start = t0.clone()
            for _ in range(3):
                drv.step()
            drv.restart(t_start, t_start[part.lo:part.hi])
        norms = []
        for s in range(steps):
            drv.step()
            if every == 1:
                norms.append(drv.term_norm())
            elif (s + 1) % every == 0 or s + 1 == steps:       # batched log: one all-reduce per `every` steps (and one for the tail)
                norms.extend(drv.term_norms_of_batch())
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), x=x.numpy(), t=drv.term.numpy()[part.lo:part.hi],
                 norms=np.asarray(norms), lo=part.lo, hi=part.hi, sent=ex.bytes_sent_per_step())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,k,w,split", [(4000, 8, 0, False), (4001, 8, 0, True), (6000, 12, 300, False), (6000, 12, 300, True)])
def test_partitioned_iteration_matches_single_process(tmp_path, n, k, w, split):
    world, steps = 2, 4
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, k, w, steps, str(tmp_path), split), nprocs=world, join=True)
    rp, ci, va, b = G.sdd_rows(n, k, 3, w)
    o = O.neumann_solve(rp, ci, va, b, max_terms=steps + 1, series_tolerance=0.0, max_iterations=steps + 1, tolerance=0.0)
    xs, ts = np.zeros(n), np.zeros(n)
    for r in range(world):
        z = np.load(tmp_path / f"rank{r}.npz")
        xs[int(z["lo"]):int(z["hi"])] = z["x"]
        ts[int(z["lo"]):int(z["hi"])] = z["t"]
        np.testing.assert_allclose(z["norms"], o["term_norms"][1:], rtol=1e-12)
        if w:
            assert int(z["sent"]) == 8 * w          # one neighbour each at world = 2
    assert (xs.view(np.uint64) == o["x"].view(np.uint64)).all()
    assert (ts.view(np.uint64) == o["term"].view(np.uint64)).all()


@pytest.mark.parametrize("split", [False, True])
def test_halo_as_one_allreduce(tmp_path, monkeypatch, split):
    """the halo moved as ONE all-reduce over a zero-filled compact strip buffer (north_star's wording): same bits, three ranks"""
    monkeypatch.setenv("SL_TEST_EXCHANGE", "allreduce")
    n, k, w, world, steps = 6000, 12, 300, 3, 4
    mp.spawn(_worker, args=(world, _free_port(), n, k, w, steps, str(tmp_path), split), nprocs=world, join=True)
    rp, ci, va, b = G.sdd_rows(n, k, 3, w)
    o = O.neumann_solve(rp, ci, va, b, max_terms=steps + 1, series_tolerance=0.0, max_iterations=steps + 1, tolerance=0.0)
    xs, ts = np.zeros(n), np.zeros(n)
    for r in range(world):
        z = np.load(tmp_path / f"rank{r}.npz")
        xs[int(z["lo"]):int(z["hi"])] = z["x"]
        ts[int(z["lo"]):int(z["hi"])] = z["t"]
        assert int(z["sent"]) == 8 * 2 * w * (world - 1)
    assert (xs.view(np.uint64) == o["x"].view(np.uint64)).all()
    assert (ts.view(np.uint64) == o["term"].view(np.uint64)).all()


@pytest.mark.parametrize("every", ["1", "4"])
def test_restart_puts_the_driver_back_at_the_start(tmp_path, monkeypatch, every):
    """PartitionedNeumann.restart (bench.py's parity gate on the torch.distributed path runs steps, reads them and starts over): the
    iteration after a restart is the iteration of a fresh driver — bits of x and t, every norm — with the split step and a batched log"""
    monkeypatch.setenv("SL_TEST_RESTART", "1")
    monkeypatch.setenv("SL_TEST_REDUCE_EVERY", every)
    n, k, w, world, steps = 6000, 12, 300, 2, 6
    mp.spawn(_worker, args=(world, _free_port(), n, k, w, steps, str(tmp_path), True), nprocs=world, join=True)
    rp, ci, va, b = G.sdd_rows(n, k, 3, w)
    o = O.neumann_solve(rp, ci, va, b, max_terms=steps + 1, series_tolerance=0.0, max_iterations=steps + 1, tolerance=0.0)
    xs, ts = np.zeros(n), np.zeros(n)
    for r in range(world):
        z = np.load(tmp_path / f"rank{r}.npz")
        xs[int(z["lo"]):int(z["hi"])] = z["x"]
        ts[int(z["lo"]):int(z["hi"])] = z["t"]
        assert len(z["norms"]) == steps
        np.testing.assert_allclose(z["norms"], o["term_norms"][1:], rtol=1e-12)
    assert (xs.view(np.uint64) == o["x"].view(np.uint64)).all() and (ts.view(np.uint64) == o["term"].view(np.uint64)).all()


def test_partitioned_iteration_batched_norm_log(tmp_path, monkeypatch):
    """reduce_every = 4 over 10 steps: two full logs and a tail of two — every step's norm still arrives globally summed"""
    monkeypatch.setenv("SL_TEST_REDUCE_EVERY", "4")
    n, k, w, world, steps = 6000, 12, 300, 2, 10
    mp.spawn(_worker, args=(world, _free_port(), n, k, w, steps, str(tmp_path), True), nprocs=world, join=True)
    rp, ci, va, b = G.sdd_rows(n, k, 3, w)
    o = O.neumann_solve(rp, ci, va, b, max_terms=steps + 1, series_tolerance=0.0, max_iterations=steps + 1, tolerance=0.0)
    for r in range(world):
        z = np.load(tmp_path / f"rank{r}.npz")
        assert len(z["norms"]) == steps
        np.testing.assert_allclose(z["norms"], o["term_norms"][1:], rtol=1e-12)


def _solve_worker(rank, world, port, n, k, w, tol, scaled, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        part = D.RowPartition(n, world, rank)
        rp, ci, va, b = G.sdd_rows(n, k, 3, w, part.lo, part.hi)
        d = np.array([va[i * k:(i + 1) * k][ci[i * k:(i + 1) * k] == part.lo + i][0] for i in range(part.n_local)])
        dinv = 1.0 / d

        def residual_norm2(x_full, rhs_local, out):
            r = rhs_local.numpy() - O.spmv(rp, ci, va, x_full.numpy())     # rhs_local is b, or D^-1 b for the reference's scaled form
            out[0] = float(np.dot(r, r))

        def axpy(alpha, xv, yv):
            yv.add_(xv, alpha=alpha)

        def sumsq(v, out):
            out[0] = float(np.dot(v.numpy(), v.numpy()))

        ops = D.LocalOps(_oracle_local_step(rp, ci, va, dinv, part.lo), residual_norm2, axpy, sumsq)
        ex = D.AllGatherExchange(part) if w == 0 else D.HaloExchange(part, w)
        s = D.PartitionedNeumannSolver(part, ops, ex, reference_scaled_residual=scaled)
        r = s.solve(torch.from_numpy(b), torch.from_numpy(dinv), tolerance=tol)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), x=r.solution_local.numpy(), lo=part.lo, hi=part.hi, it=r.iterations,
                 terms=r.terms_computed, conv=r.converged, resn=r.residual_norm, norms=np.asarray(r.term_norms))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,k,w,tol,scaled", [(4000, 8, 0, 1e-10, False), (6000, 12, 300, 1e-6, False), (6000, 12, 300, 1e-12, True)])
def test_partitioned_solve_matches_single_process(tmp_path, n, k, w, tol, scaled):
    """the whole NeumannSolver::solve loop over 2 ranks: same iteration count, stop reason and bits as one process"""
    world = 2
    mp.spawn(_solve_worker, args=(world, _free_port(), n, k, w, tol, scaled, str(tmp_path)), nprocs=world, join=True)
    rp, ci, va, b = G.sdd_rows(n, k, 3, w)
    o = O.neumann_solve(rp, ci, va, b, tolerance=tol, residual=1 if scaled else 0)
    xs = np.zeros(n)
    for r in range(world):
        z = np.load(tmp_path / f"rank{r}.npz")
        xs[int(z["lo"]):int(z["hi"])] = z["x"]
        assert int(z["it"]) == o["iterations"] and int(z["terms"]) == o["terms"] and bool(z["conv"]) == o["converged"]
        np.testing.assert_allclose(z["norms"], o["term_norms"][:int(z["terms"])], rtol=1e-12)
        assert abs(float(z["resn"]) - o["residual_norm"]) <= 1e-12 * max(1.0, o["residual_norm"])
    assert (xs.view(np.uint64) == o["x"].view(np.uint64)).all()


def test_split_bounds():
    assert D.split_bounds(1000, 100, True, True) == ([(0, 100), (900, 1000)], [(100, 900)])
    assert D.split_bounds(1000, 100, False, True) == ([(0, 0), (900, 1000)], [(0, 900)])
    assert D.split_bounds(1000, 100, True, False) == ([(0, 100), (1000, 1000)], [(100, 1000)])
    assert D.split_bounds(150, 100, True, True) == ([(0, 100), (100, 150)], [(100, 100)])


def _ragged_system(n):
    """row dominant with power-law row lengths: the transpose of the PageRank system, I - 0.85 P (rows = out-links, Zipf degrees)"""
    import scipy.sparse as sp
    rp, ci, va, b = G.pagerank_system(n, *G.pagerank_graph(n, 5))
    T = sp.csr_matrix((va, ci.astype(np.int64), rp.astype(np.int64)), shape=(n, n)).T.tocsr()
    T.sort_indices()
    return T.indptr.astype(np.uint32), T.indices.astype(np.uint32), T.data.astype(np.float64), 1.0 + 0.001 * (np.arange(n) % 1000)


def _ragged_worker(rank, world, port, n, steps, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rp, ci, va, b = _ragged_system(n)
        part = D.RowPartition(n, world, rank, bounds=D.nnz_balanced_bounds(rp, world))
        lo, hi = part.lo, part.hi
        prp = (rp[lo:hi + 1].astype(np.int64) - int(rp[lo])).astype(np.uint32)
        pci, pva = ci[rp[lo]:rp[hi]], va[rp[lo]:rp[hi]]
        dinv_all = 1.0 / np.array([va[rp[i]:rp[i + 1]][ci[rp[i]:rp[i + 1]] == i][0] for i in range(n)])
        t0 = torch.from_numpy(b * dinv_all)
        x = t0[lo:hi].clone()
        drv = D.PartitionedNeumann(part, _oracle_local_step(prp, pci, pva, dinv_all[lo:hi], lo), D.AllGatherExchange(part), t0, x)
        norms = []
        for _ in range(steps):
            drv.step()
            norms.append(drv.term_norm())
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), x=x.numpy(), t=drv.term.numpy()[lo:hi], norms=np.asarray(norms), lo=lo, hi=hi,
                 nnz=int(rp[hi]) - int(rp[lo]))
    finally:
        dist.destroy_process_group()


def test_nnz_balanced_partition_of_ragged_rows(tmp_path):
    """row ranges balanced by stored entries (prefix sum of row_ptr).  Three ranks with unequal row counts,
    all-gather over unequal chunks; per-row bits equal the single-process iteration."""
    n, world, steps = 3000, 3, 4
    mp.spawn(_ragged_worker, args=(world, _free_port(), n, steps, str(tmp_path)), nprocs=world, join=True)
    rp, ci, va, b = _ragged_system(n)
    o = O.neumann_solve(rp, ci, va, b, max_terms=steps + 1, series_tolerance=0.0, max_iterations=steps + 1, tolerance=0.0)
    xs, ts, sizes, nnzs = np.zeros(n), np.zeros(n), [], []
    for r in range(world):
        z = np.load(tmp_path / f"rank{r}.npz")
        xs[int(z["lo"]):int(z["hi"])] = z["x"]
        ts[int(z["lo"]):int(z["hi"])] = z["t"]
        sizes.append(int(z["hi"]) - int(z["lo"]))
        nnzs.append(int(z["nnz"]))
        np.testing.assert_allclose(z["norms"], o["term_norms"][1:], rtol=1e-12)
    assert sum(sizes) == n and max(sizes) > 1.1 * min(sizes)             # the row counts really differ ...
    assert max(nnzs) - min(nnzs) <= 2 * int(np.diff(rp.astype(np.int64)).max())   # ... because the entries are balanced
    assert (xs.view(np.uint64) == o["x"].view(np.uint64)).all()
    assert (ts.view(np.uint64) == o["term"].view(np.uint64)).all()


def test_row_partition_bounds():
    b = D.nnz_balanced_bounds([0, 10, 10, 11, 12, 20], 2)
    assert b == [0, 1, 5]
    q = D.RowPartition(5, 2, 1, bounds=b)
    assert (q.lo, q.hi, q.n_padded, q.uniform) == (1, 5, 5, False) and q.range_of(0) == (0, 1)
    with pytest.raises(ValueError):
        D.RowPartition(5, 2, 0, bounds=[0, 3, 4])
    p = [D.RowPartition(10, 4, r) for r in range(4)]
    assert [(q.lo, q.hi) for q in p] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert p[0].n_padded == 12
    with pytest.raises(ValueError):
        D.HaloExchange(D.RowPartition(100, 4, 1), 30)


def test_halo_exchanges_reject_partitions_with_an_empty_rank():
    """n = 9 over 4 ranks: the uniform partition gives the last rank [9, 9) — its neighbour would skip it while it posts
    sends (unmatched ops, hang); an empty MIDDLE rank of explicit bounds would ship stale strips.  Both are refused."""
    assert D.RowPartition(9, 4, 3).n_local == 0
    for cls in (D.HaloExchange, D.HaloAllReduceExchange):
        for r in range(4):
            with pytest.raises(ValueError, match="rows on every rank"):
                cls(D.RowPartition(9, 4, r), 1)
        with pytest.raises(ValueError, match="rows on every rank"):
            cls(D.RowPartition(12, 3, 0, bounds=[0, 6, 6, 12]), 2)
        cls(D.RowPartition(12, 3, 1, bounds=[0, 4, 8, 12]), 2)      # fine


def test_bench_launcher_without_a_gpu_fails_loudly_after_trying_every_transport():
    """`python bench.py --gpus 2` with no launcher around it: the parent starts two ranks per attempt; without a device every rank stops
    with the library's loud DeviceError (no CPU fallback), the parent walks the transports in order and exits non-zero with the list
    of attempts — the launcher's plumbing (spawn, collect, fall back, give up) without a GPU"""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the GPU suite runs the launcher for real")
    env = dict(os.environ, SL_BENCH_TRANSPORTS="ipc,rccl", SL_COMM_TIMEOUT_MS="3000")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--rows", "20000", "--steps", "2", "--warmup", "0", "--attempt-timeout", "120"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=str(root))
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], r.stdout[-500:]
    assert "attempt 0 (transport ipc) failed" in r.stderr and "attempt 1 (transport rccl) failed" in r.stderr and "every transport failed" in r.stderr, r.stderr[-3000:]
