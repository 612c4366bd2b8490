"""The partitioned NeumannSolver::solve driver (distributed.PartitionedNeumannSolver) on one GPU: with world = 1 the
exchanges are no-ops, so it must reproduce sl_neumann_solve (and the oracle) exactly — same iteration count, same
stop reason, same solution bits.  The 2-rank exchange logic is covered on CPU (tests/test_distributed_cpu.py)."""
import numpy as np
import pytest

import sublinear_time_solver_amd as S
from sublinear_time_solver_amd import distributed as D
from sublinear_time_solver_amd import generators as G
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,k,w,tol,order,scaled", [(30_011, 16, 700, 1e-10, 0, False), (30_011, 13, 0, 1e-7, 1, False),
                                                    (20_000, 8, 300, 1e-12, 0, True)])
def test_partitioned_solver_world1_equals_single_gpu_solve(gpu, n, k, w, tol, order, scaled):
    import torch
    rp, ci, va, b = G.sdd_rows(n, k, seed=3, half_bandwidth=w)
    m = S.SparseMatrix.from_csr(rp, ci, va, n, n)
    part = D.RowPartition(n, 1, 0)
    dinv = torch.from_numpy(m.diagonal_inverse()).cuda()
    ops = D.hip_local_ops(m._h, dinv, order)
    ex = D.HaloExchange(part, w) if w else D.AllGatherExchange(part)
    r = D.PartitionedNeumannSolver(part, ops, ex, reference_scaled_residual=scaled).solve(torch.from_numpy(b).cuda(), dinv, tolerance=tol)
    g = S.NeumannSolver(order=order, residual=1 if scaled else 0).solve(m, b, S.SolverOptions(tolerance=tol))
    o = O.neumann_solve(rp, ci, va, b, tolerance=tol, order=order, residual=1 if scaled else 0)
    x = r.solution_local.cpu().numpy()
    assert r.iterations == g.iterations == o["iterations"] and r.converged == g.converged == o["converged"]
    assert (x.view(np.uint64) == np.ascontiguousarray(g.solution).view(np.uint64)).all()
    assert (x.view(np.uint64) == o["x"].view(np.uint64)).all()
    assert abs(r.residual_norm - g.residual_norm) <= 1e-12 * max(1.0, g.residual_norm)


def test_step_partials_pieces_equal_the_whole_step(gpu):
    """sl_neumann_step_partials + sl_reduce_partials over three row-slice matrices (boundary / interior / boundary) = one
    sl_neumann_step over the whole matrix: t and x bit for bit, the norm to rounding (different partial-sum grouping)"""
    import ctypes as C
    import torch
    from sublinear_time_solver_amd import _lib as L
    lib = L.load()
    n, k, w = 50_000, 16, 900
    rp, ci, va, b = G.sdd_rows(n, k, seed=4, half_bandwidth=w)
    whole = S.SparseMatrix.from_csr(rp, ci, va, n, n)
    dinv = torch.from_numpy(whole.diagonal_inverse()).cuda()
    t_in = torch.from_numpy(np.cos(np.arange(n) * 0.01)).cuda()
    x0 = torch.from_numpy(np.sin(np.arange(n) * 0.02)).cuda()
    t_ref, x_ref, nrm_ref = torch.empty(n, dtype=torch.float64, device="cuda"), x0.clone(), torch.zeros(2, dtype=torch.float64, device="cuda")
    L.check(lib.sl_neumann_step(whole._h, dinv.data_ptr(), t_in.data_ptr(), t_ref.data_ptr(), x_ref.data_ptr(), nrm_ref.data_ptr(), 0))
    bounds = [(0, w), (w, n - w), (n - w, n)]
    pieces = [S.SparseMatrix.from_csr((rp[lo:hi + 1] - rp[lo]).astype(np.uint32), ci[rp[lo]:rp[hi]], va[rp[lo]:rp[hi]], hi - lo, n, row_offset=lo)
              for lo, hi in bounds]
    cap = 0
    for p in pieces:
        c = L.u64(0)
        L.check(lib.sl_matrix_partials_capacity(p._h, C.byref(c)))
        cap += c.value
    partials = torch.zeros(cap, dtype=torch.float64, device="cuda")
    t_out, x = torch.empty(n, dtype=torch.float64, device="cuda"), x0.clone()
    used = 0
    for (lo, hi), p in zip(bounds, pieces):
        got = C.c_uint32(0)
        L.check(lib.sl_neumann_step_partials(p._h, dinv[lo:hi].data_ptr(), t_in.data_ptr(), t_out[lo:hi].data_ptr(), x[lo:hi].data_ptr(),
                                             partials[used:].data_ptr(), C.byref(got), 0))
        used += got.value
    assert 0 < used <= cap
    nrm = torch.zeros(2, dtype=torch.float64, device="cuda")
    L.check(lib.sl_reduce_partials(partials.data_ptr(), used, nrm.data_ptr()))
    L.check(lib.sl_synchronize())
    assert torch.equal(t_out.view(torch.int64), t_ref.view(torch.int64)) and torch.equal(x.view(torch.int64), x_ref.view(torch.int64))
    assert abs(float(nrm[0]) - float(nrm_ref[0])) <= 1e-12 * float(nrm_ref[0])
