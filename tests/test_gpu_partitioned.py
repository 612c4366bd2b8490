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
