"""a14 / a13 on the GPU (pytest -m gpu).

  * sl_forward_push_southwell = TS solveForwardPush (src/core/solver.ts:437-522) in the reference's own visiting order: G6 of
    SURVEY §8(c) (10 x 10 tridiag(-1, 10, -1), b = e_0 + e_9, epsilon 1e-3 -> 12 pushes, ||r|| = 5.2915e-4,
    tests/mcp/mcp-tool-tests.js:27-52), then push sequence / solution / residual bit for bit against the oracle restatement on
    seeded systems incl. duplicate column entries, ties, the failure paths.
  * sl_push_options.theta_rows = the degree-scaled admission rule of the orphan Rust spec (forward_push.rs:93-99): bitwise against
    the oracle's synchronous push with the same per-row thresholds, and the spec's exit condition on a skewed graph."""
import numpy as np
import pytest

import sublinear_time_solver_amd as S
from sublinear_time_solver_amd import generators as G
from sublinear_time_solver_amd.push_graph import ForwardPushConfig, ForwardPushSolver, PushGraph
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def _tridiag10():
    tr, tc, tv = [], [], []
    for i in range(10):
        for j, v in ((i - 1, -1.0), (i, 10.0), (i + 1, -1.0)):
            if 0 <= j < 10:
                tr.append(i), tc.append(j), tv.append(v)
    return O.csr_from_triplets(tr, tc, tv, 10, 10)


def test_g6_ts_forward_push_on_the_gpu(gpu):
    rp, ci, va = _tridiag10()
    b = np.zeros(10)
    b[0] = b[9] = 1.0
    m = S.SparseMatrix.from_csr(rp, ci, va, 10, 10, with_transpose=True)
    g = S.GaussSouthwellSolver(epsilon=1e-3, max_iterations=1000).solve(m, b, log_pushes=True)
    assert g["converged"] and g["iterations"] == 12
    assert abs(g["residual"] - 5.2915e-4) < 1e-7
    o = O.ts_forward_push(rp, ci, va, b, 1e-3, 1000)
    assert (bits(g["solution"]) == bits(o["x"])).all() and (bits(g["residual_vector"]) == bits(o["r"])).all()
    assert g["push_log"][:2].tolist() == [0, 9]                       # first maximum first: index 0 before index 9 (equal residuals)
    # the TS surface: method 'forward-push' reports pushes as iterations, like the reference
    r = S.SublinearSolver(method="forward-push", epsilon=1e-3, max_iterations=1000).solve(
        {"rows": 10, "cols": 10, "format": "coo", "values": va.tolist(), "rowIndices": np.repeat(np.arange(10), np.diff(rp)).tolist(),
         "colIndices": ci.tolist()}, b.tolist())
    assert r["iterations"] == 12 and r["converged"] and abs(r["residual"] - 5.2915e-4) < 1e-7


def _oracle_sequence(rp, ci, va, b, eps, max_it):
    """push order of the reference loop, from the oracle state after k = 1, 2, ... pushes (x changes in exactly one place)"""
    seq, prev = [], np.zeros(b.size)
    for k in range(1, max_it + 1):
        o = O.ts_forward_push(rp, ci, va, b, eps, k)
        if o["iterations"] < k:
            break
        ch = np.nonzero(bits(o["x"]) != bits(prev))[0]
        seq.append(int(ch[0]) if ch.size else -1)
        prev = o["x"].copy()
    return seq


@pytest.mark.parametrize("n,k,seed", [(300, 8, 2), (2000, 5, 7)])
def test_push_sequence_solution_and_residual_match_the_reference_order(gpu, n, k, seed):
    rp, ci, va, b = G.sdd_rows(n, k, seed)
    b = b * (np.arange(n) % 7 == 0)
    m = S.SparseMatrix.from_csr(rp, ci, va, n, n, with_transpose=True)
    eps = 1e-5
    o = O.ts_forward_push(rp, ci, va, b, eps, 200_000)
    g = S.GaussSouthwellSolver(epsilon=eps, max_iterations=200_000).solve(m, b, log_pushes=True)
    assert o["status"] == 0 and g["converged"] and g["iterations"] == o["iterations"] > n // 7
    assert (bits(g["solution"]) == bits(o["x"])).all() and (bits(g["residual_vector"]) == bits(o["r"])).all()
    assert abs(g["residual"] - o["residual"]) <= 1e-12 * max(1.0, o["residual"])
    if n <= 300:
        seq = _oracle_sequence(rp, ci, va, b, eps, 40)
        assert g["push_log"][:len(seq)].tolist() == seq
    # the invariant r = b - A x
    np.testing.assert_allclose(b - O.spmv(rp, ci, va, g["solution"]), g["residual_vector"], atol=1e-12)


def test_duplicate_column_entries_ties_and_failure_paths(gpu):
    # column 1 holds row 2 twice (stored duplicates are kept, matrix/sparse.rs:80-132): the two updates of r_2 happen in stored order
    tr = [0, 0, 1, 1, 2, 2, 2, 2, 3, 3]
    tc = [0, 1, 1, 2, 1, 1, 2, 3, 3, 0]
    tv = [4.0, 1.0, 5.0, -1.0, 0.3, 0.7, 6.0, 1.0, 3.0, -1.0]
    rp, ci, va = O.csr_from_triplets(tr, tc, tv, 4, 4)
    b = np.array([1.0, -2.0, 2.0, 1.0])                                # |r_1| = |r_2|: the first maximum (index 1) goes first
    m = S.SparseMatrix.from_csr(rp, ci, va, 4, 4, with_transpose=True)
    o = O.ts_forward_push(rp, ci, va, b, 1e-12, 10_000)
    g = S.GaussSouthwellSolver(epsilon=1e-12, max_iterations=10_000).solve(m, b, log_pushes=True)
    assert g["push_log"][0] == 1 and g["iterations"] == o["iterations"]
    assert (bits(g["solution"]) == bits(o["x"])).all() and (bits(g["residual_vector"]) == bits(o["r"])).all()
    # max_iterations reached: ConvergenceFailure like solver.ts:509-515; the partial state equals the oracle's after as many pushes
    with pytest.raises(S.SolverError) as e:
        S.GaussSouthwellSolver(epsilon=1e-12, max_iterations=3).solve(m, b)
    assert e.value.kind == "ConvergenceFailure"
    part = S.GaussSouthwellSolver(epsilon=1e-12, max_iterations=3).solve(m, b, on_failure="return")
    o3 = O.ts_forward_push(rp, ci, va, b, 1e-12, 3)
    assert part["iterations"] == 3 and not part["converged"] and (bits(part["solution"]) == bits(o3["x"])).all()
    # zero right-hand side: converged at once, zero pushes, residual = Infinity as in the reference (solver.ts:446)
    z = S.GaussSouthwellSolver(epsilon=1e-6).solve(m, np.zeros(4))
    assert z["converged"] and z["iterations"] == 0 and np.isinf(z["residual"]) and z["solution"].sum() == 0.0
    # a zero diagonal under the largest residual
    rp2, ci2, va2 = O.csr_from_triplets([0, 0, 1], [0, 1, 0], [1e-16, 1.0, 1.0], 2, 2)
    m2 = S.SparseMatrix.from_csr(rp2, ci2, va2, 2, 2, with_transpose=True)
    with pytest.raises(S.SolverError) as e:
        S.GaussSouthwellSolver(epsilon=1e-6).solve(m2, np.array([1.0, 0.5]))
    assert e.value.kind == "NumericalInstability" and "Zero diagonal at position 0" in str(e.value)


# ---- a13: degree-scaled admission rule -------------------------------------------------------------------------------------
def test_per_row_thresholds_bitwise_against_the_oracle(gpu):
    n, k = 6000, 9
    rp, ci, va, b = G.sdd_rows(n, k, seed=13)
    b = b * (np.arange(n) % 3 == 0)
    th = 1e-7 * (1.0 + (np.arange(n) % 17) ** 2)                        # thresholds over a factor of 257
    m = S.SparseMatrix.from_csr(rp, ci, va, n, n, with_transpose=True)
    for dense_switch in (1.0 / 16.0, 1e-9):                              # sparse rounds / dense rounds from the start
        q = O.push_sync_solve(rp, ci, va, b, theta=1.0, theta_rows=th, log_cap=1 << 22)
        p = S.PushSolver(theta=1.0, theta_rows=th, dense_switch=dense_switch).solve(m, b, log_frontier=1 << 22)
        assert p["converged"] and p["rounds"] == q["rounds"] and p["pushes"] == q["pushes"]
        assert (p["frontier_log"] == q["frontier_log"]).all()
        assert (bits(p["solution"]) == bits(q["x"])).all() and (bits(p["residual"]) == bits(q["r"])).all()
    dinv = 1.0 / np.array([va[rp[i]:rp[i + 1]][ci[rp[i]:rp[i + 1]] == i][0] for i in range(n)])
    assert (np.abs(p["residual"] * dinv) < th).all()                    # every row below ITS threshold at exit


def test_degree_scaled_rule_on_a_skewed_graph(gpu):
    """a hub with 300 out-edges among nodes of out-degree 2: the spec pushes node u while residual[u] >= epsilon * max(deg_u, 1)
    (forward_push.rs:93-99) — at exit every node is below that, pushes are fewer than with the absolute threshold, and the
    estimates stay within the spec's own error of the sequential restatement"""
    rng = np.random.default_rng(5)
    n, edges = 400, []
    for i in range(n):
        outs = rng.choice(n, size=300 if i == 0 else 2, replace=False)
        for t in outs:
            if t != i:
                edges.append((i, int(t), float(rng.uniform(0.5, 2.0))))
    g = PushGraph.from_edges(n, edges)
    eps = 1e-6
    scaled = ForwardPushSolver(g, ForwardPushConfig(alpha=0.15, epsilon=eps)).solve_single_source(0)
    absolute = ForwardPushSolver(g, ForwardPushConfig(alpha=0.15, epsilon=eps, degree_scaled=False)).solve_single_source(0)
    deg = np.maximum(g.degrees, 1.0)
    assert (scaled.residual < eps * deg).all()                          # the spec's exit condition
    assert not (scaled.residual < eps).all() or g.degrees.max() <= 1.0  # ... which is weaker than the absolute one on this graph
    assert (absolute.residual < eps).all()
    assert scaled.push_count < absolute.push_count
    for r in (scaled, absolute):
        assert (r.estimate >= 0).all() and abs(r.estimate.sum() + r.residual.sum() - 1.0) < 1e-12
    acl = O.acl_push(g.row_ptr, g.col_idx, g.weights, [0], alpha=0.15, epsilon=eps)
    # both stop with sum(residual) <= epsilon * sum(deg): the estimates agree within that mass
    bound = eps * deg.sum()
    assert np.abs(scaled.estimate - acl["estimate"]).max() <= bound
    assert (acl["residual"] < eps * deg + 1e-18).all()


def test_g12_the_references_own_typescript_push_on_the_gpu(gpu):
    """tests/golden/reference_ts_push.npz holds what the reference's solveForwardPush — its own TypeScript, executed on these inputs by
    tests/golden/make_golden_ts_push.py — returned: the device's solution must carry the same bits after the same number of pushes (ties of
    |r_i| on a PageRank-type system, an asymmetric tridiagonal, 10^3 pushes on a band, and the CONVERGENCE_FAILED exit after maxIterations)"""
    from pathlib import Path
    g = np.load(Path(__file__).resolve().parent / "golden" / "reference_ts_push.npz")
    seen_failure = False
    for k in map(str, g["names"]):
        n, maxit, conv, its = (int(v) for v in g[k + "/params"])
        eps, res = (float(v) for v in g[k + "/epsilon_residual"])
        rp, ci, va = O.csr_from_triplets(g[k + "/rows"], g[k + "/cols"], g[k + "/values"], n, n)
        m = S.SparseMatrix.from_csr(rp, ci, va, n, n, with_transpose=True)
        out = S.GaussSouthwellSolver(epsilon=eps, max_iterations=maxit).solve(m, g[k + "/b"], on_failure="return")
        assert out["iterations"] == its and out["converged"] == bool(conv), (k, out["iterations"], its)
        assert abs(out["residual"] - res) <= 1e-12 * res, (k, out["residual"], res)
        if conv:
            assert (bits(out["solution"]) == bits(g[k + "/solution"])).all(), k
        else:
            seen_failure = True
            with pytest.raises(S.SolverError) as e:                     # the reference throws there; so does the default surface
                S.GaussSouthwellSolver(epsilon=eps, max_iterations=maxit).solve(m, g[k + "/b"])
            assert e.value.kind == "ConvergenceFailure"
    assert seen_failure
