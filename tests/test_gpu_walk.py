"""Monte-Carlo estimateEntry branch: GPU walks vs the oracle restatement of the same
per-walk-stream rule — per-walk values bit-identical, mean / variance to 1e-12."""
import ctypes as C

import numpy as np
import pytest

import sublinear_time_solver_amd as S
from sublinear_time_solver_amd import _lib as L
from sublinear_time_solver_amd import generators as G
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _walk(m, b, row, n_samples, seed, eps=0.1):
    lib = L.load()
    vals = np.zeros(max(n_samples, 1))
    res = L.WalkResult()
    L.check(lib.sl_estimate_entry_random_walk(m._h, L.ptr(np.ascontiguousarray(b, dtype=np.float64)), 0, row, eps, seed, n_samples,
                                              L.ptr(vals) if n_samples else None, C.byref(res)))
    return vals, res


@pytest.mark.parametrize("n,k,w,seed", [(2000, 8, 0, 42), (5000, 16, 50, 1), (300, 5, 0, 12345)])
def test_walk_values_bitwise_vs_oracle(gpu, n, k, w, seed):
    rp, ci, va, b = G.sdd_rows(n, k, seed=3, half_bandwidth=w)
    m = S.SparseMatrix.from_csr(rp, ci, va, n, n, keep_csr=True)
    N = 20000
    for row in (0, n // 2):
        gv, res = _walk(m, b, row, N, seed)
        ov, om, ovar = O.ts_random_walk_streams(rp, ci, va, b, row, N, seed)
        assert (gv.view(np.uint64) == ov.view(np.uint64)).all(), "per-walk values must be bit-identical"
        assert res.num_samples == N
        assert abs(res.estimate - om) <= 1e-12 * max(1.0, abs(om))
        assert abs(res.variance - ovar) <= 1e-12 * max(ovar, om * om)      # both are sums of (v - mean)^2: compare on the scale of mean^2


def test_walk_sample_count_rule_and_ts_surface(gpu):
    tr, tc, tv = [], [], []
    for i in range(10):
        for j, v in ((i - 1, -1.0), (i, 10.0), (i + 1, -1.0)):
            if 0 <= j < 10:
                tr.append(i), tc.append(j), tv.append(v)
    m = S.SparseMatrix.from_triplets(zip(tr, tc, tv), 10, 10, with_transpose=True)
    _, res = _walk(m, np.ones(10), 0, 0, 42, eps=0.5)
    assert res.num_samples == 100                                     # max(100, ceil(1/eps^2)), solver.ts:586
    _, res = _walk(m, np.ones(10), 0, 0, 42, eps=0.01)
    assert res.num_samples == 10000
    out = S.SublinearSolver(method="random-walk", epsilon=0.05, seed=42).estimate_entry(m, np.ones(10), row=0, method="random-walk")
    assert set(out) >= {"estimate", "variance", "confidence"} and out["numSamples"] == 400
    assert abs(out["estimate"] - 0.1) < 1e-9                          # the reference estimator's value on this system (b_i / a_ii)
    bad = S.SparseMatrix.from_triplets([(0, 0, 1.0), (0, 1, 0.5), (1, 0, 0.5)], 2, 2, keep_csr=True)
    with pytest.raises(S.SolverError) as e:
        _walk(bad, np.ones(2), 0, 10, 1)
    assert e.value.kind == "NumericalInstability"                     # Zero diagonal (solver.ts:368-371)
