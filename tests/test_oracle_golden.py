"""Pin the CPU oracle (oracle/sl_oracle.c) against the reference's own known answers and against
outputs captured from the reference's runnable Python / JS solvers.
CPU only — runs in the build container and on the GPU box alike (nothing reads /root/reference)."""
from pathlib import Path

import numpy as np
import pytest

from oracle import oracle as O

GOLD = np.load(Path(__file__).parent / "golden" / "reference_jacobi.npz")
CASES = [str(c) for c in GOLD["__cases"]]


def _mat(case):
    key = case.rsplit("__", 1)[0]
    return GOLD[f"{key}__row_ptr"], GOLD[f"{key}__col_idx"], GOLD[f"{key}__values"]


# ---- G1: Python Jacobi iterates == Neumann partial sums --------------------------------------
@pytest.mark.parametrize("case", CASES)
def test_g1_partial_sums_match_python_jacobi(case):
    rp, ci, va = _mat(case)
    b = GOLD[f"{case}__b"]
    for k in (1, 2, 5, 10):
        r = O.neumann_solve(rp, ci, va, b, max_terms=k, series_tolerance=0.0, max_iterations=k, tolerance=0.0)
        assert r["terms"] == k
        np.testing.assert_allclose(r["x"], GOLD[f"{case}__x_k{k}"], rtol=0, atol=1e-13)
    r = O.neumann_solve(rp, ci, va, b, max_terms=2000, series_tolerance=1e-15, max_iterations=2000, tolerance=1e-12)
    assert r["converged"]
    np.testing.assert_allclose(r["x"], GOLD[f"{case}__x_final"], rtol=0, atol=1e-11)


# ---- G2: JS Jacobi final solution ---------------------------------------------------------------
@pytest.mark.parametrize("case", CASES)
def test_g2_matches_js_jacobi(case):
    rp, ci, va = _mat(case)
    b = GOLD[f"{case}__b"]
    r = O.neumann_solve(rp, ci, va, b, max_terms=2000, series_tolerance=1e-15, max_iterations=2000, tolerance=1e-13)
    xj = GOLD[f"{case}__js_x"]
    assert np.max(np.abs(r["x"] - xj)) <= 1e-9 * max(1.0, np.max(np.abs(xj)))
    # JS stops on ||b - Ax|| / ||b|| < 1e-10 (convergence-detector.js:64-72,165-172)
    res = np.linalg.norm(b - O.spmv(rp, ci, va, xj)) / np.linalg.norm(b)
    assert res < 1e-9


def test_negative_fixture_is_not_diagonally_dominant():
    rp, ci, va = GOLD["neg_n_100_sparse_dd__row_ptr"], GOLD["neg_n_100_sparse_dd__col_idx"], GOLD["neg_n_100_sparse_dd__values"]
    assert not O.is_diagonally_dominant(rp, ci, va)
    with pytest.raises(O.OracleError) as e:
        O.neumann_solve(rp, ci, va, np.ones(100))
    assert e.value.kind == "MatrixNotDiagonallyDominant"       # neumann.rs:163-169


# ---- G3: primitives from the reference's unit tests (exact) ------------------------------------
def test_g3_spmv_kats():
    # sparse.rs:923-933 / simd_ops.rs:259-268
    rp, ci, va = O.csr_from_triplets([0, 0, 1, 1], [0, 1, 0, 1], [2.0, 1.0, 1.0, 3.0], 2, 2)
    for order in (O.ORDER_SEQ, O.ORDER_SIMD4):
        assert O.spmv(rp, ci, va, [1.0, 2.0], order).tolist() == [4.0, 7.0]
    assert O.spmv(rp, ci, va, [1.0, 2.0], threads=2).tolist() == [4.0, 7.0]
    # fast_solver.rs:260-272
    rp, ci, va = O.csr_from_triplets([0, 0, 1, 1], [0, 1, 0, 1], [4.0, 1.0, 2.0, 3.0], 2, 2)
    assert O.spmv(rp, ci, va, [1.0, 2.0]).tolist() == [6.0, 8.0]
    # optimized_solver.rs:389-396
    rp, ci, va = O.csr_from_triplets([0, 0, 1, 1], [0, 1, 0, 1], [4.0, 1.0, 1.0, 3.0], 2, 2)
    assert O.spmv(rp, ci, va, [1.0, 2.0]).tolist() == [6.0, 7.0]


def test_g3_dot_axpy_norms():
    assert O.dot_simd4([1, 2, 3, 4], [5, 6, 7, 8]) == 70.0                       # simd_ops.rs:270-276
    assert O.dot_sequential([1, 2, 3, 4], [5, 6, 7, 8]) == 70.0
    assert O.axpy(2.0, [1, 2, 3, 4], [1, 1, 1, 1]).tolist() == [3.0, 5.0, 7.0, 9.0]  # simd_ops.rs:278-286
    v = [3.0, 4.0]                                                                # solver/mod.rs:588-595
    assert O.l1_norm(v) == 7.0 and O.l2_norm(v) == 5.0 and O.linf_norm(v) == 4.0


def test_g3_neumann_state_creation():
    # neumann.rs:634-648: A = [[2,1],[1,3]]... dinv = [0.5, 1/3], rhs = b * dinv = [2, 2] for b = [4, 6]
    rp, ci, va = O.csr_from_triplets([0, 0, 1, 1], [0, 1, 0, 1], [2.0, 1.0, 1.0, 3.0], 2, 2)
    dinv, rhs = O.neumann_init(rp, ci, va, [4.0, 6.0])
    assert dinv.tolist() == [0.5, 1.0 / 3.0]
    assert rhs.tolist() == [2.0, 6.0 * (1.0 / 3.0)]


def test_g3_dd_and_get():
    # matrix/mod.rs:603-613
    rp, ci, va = O.csr_from_triplets([0, 0, 1, 1], [0, 1, 0, 1], [5.0, 1.0, 2.0, 7.0], 2, 2)
    assert O.is_diagonally_dominant(rp, ci, va)
    rp2, ci2, va2 = O.csr_from_triplets([0, 0, 1, 1], [0, 1, 0, 1], [1.0, 5.0, 2.0, 1.0], 2, 2)
    assert not O.is_diagonally_dominant(rp2, ci2, va2)
    # sparse.rs:910-920: get hits / misses
    rp3, ci3, va3 = O.csr_from_triplets([0, 0, 1, 2, 2], [0, 2, 1, 0, 2], [1.0, 2.0, 3.0, 4.0, 5.0], 3, 3)
    assert O.csr_get(rp3, ci3, va3, 0, 0) == 1.0 and O.csr_get(rp3, ci3, va3, 0, 2) == 2.0
    assert O.csr_get(rp3, ci3, va3, 0, 1) is None and O.csr_get(rp3, ci3, va3, 1, 1) == 3.0


def test_a1_triplet_rules():
    # zeros dropped, stable sort by (row, col), duplicates kept separately (sparse.rs:80-132,530-548)
    rp, ci, va = O.csr_from_triplets([1, 0, 0, 1, 0], [1, 1, 0, 1, 1], [5.0, 2.0, 0.0, 6.0, 3.0], 2, 2)
    assert rp.tolist() == [0, 2, 4] and ci.tolist() == [1, 1, 1, 1] and va.tolist() == [2.0, 3.0, 5.0, 6.0]
    with pytest.raises(O.OracleError) as e:
        O.csr_from_triplets([2], [0], [1.0], 2, 2)
    assert e.value.kind == "IndexOutOfBounds"
    with pytest.raises(O.OracleError) as e:
        O.csr_from_triplets([0], [0], [float("nan")], 2, 2)
    assert e.value.kind == "InvalidInput"
    rp, ci, va = O.csr_from_triplets([], [], [], 3, 3)          # empty
    assert rp.tolist() == [0, 0, 0, 0] and ci.size == 0


def test_simd4_order_differs_from_sequential_only_in_rounding():
    rng = np.random.default_rng(7)
    n, k = 50, 13
    cols = np.stack([np.sort(rng.choice(n, k, replace=False)) for _ in range(n)])
    vals = rng.standard_normal((n, k))
    rp = np.arange(n + 1, dtype=np.uint32) * k
    x = rng.standard_normal(n)
    y0 = O.spmv(rp, cols.ravel(), vals.ravel(), x, O.ORDER_SEQ)
    y1 = O.spmv(rp, cols.ravel(), vals.ravel(), x, O.ORDER_SIMD4)
    np.testing.assert_allclose(y0, y1, rtol=1e-13, atol=1e-13)
    # hand restatement of simd_ops.rs:41-77 for one row
    v, c = vals[3], cols[3]
    lanes = [0.0] * 4
    for q in range(k // 4):
        for j in range(4):
            lanes[j] = lanes[j] + v[4 * q + j] * x[c[4 * q + j]]
    y = ((lanes[0] + lanes[1]) + lanes[2]) + lanes[3]
    for t in range(4 * (k // 4), k):
        y = y + v[t] * x[c[t]]
    assert y1[3] == y


# ---- G4: Rust-Neumann restatement KATs ------------------------------------
def test_g4_rust_neumann_quirks():
    rp, ci, va = O.csr_from_triplets([0, 0, 1, 1], [0, 1, 0, 1], [4.0, 1.0, 1.0, 3.0], 2, 2)
    b = [5.0, 4.0]
    # reference behaviour (scaled residual never passes, series criterion stops at 17 terms)
    r = O.neumann_solve(rp, ci, va, b, max_terms=20, series_tolerance=1e-8, start=O.START_ZERO,
                        residual=O.RESIDUAL_REFERENCE_SCALED)
    np.testing.assert_allclose(r["x"], [1.0, 1.0], atol=1e-7)
    assert r["iterations"] == 17 and r["converged"]
    # with the TRUE residual the every-5th-iteration check (neumann.rs:489-491) fires one term earlier
    t = O.neumann_solve(rp, ci, va, b, max_terms=20, series_tolerance=1e-8, start=O.START_ZERO)
    np.testing.assert_allclose(t["x"], [1.0, 1.0], atol=1e-7)
    assert t["iterations"] == 16 and t["converged"] and t["residual_norm"] < 1e-6
    q = O.neumann_solve(rp, ci, va, b, max_terms=20, series_tolerance=1e-8, start=O.START_REFERENCE_DEFAULT,
                        residual=O.RESIDUAL_REFERENCE_SCALED)
    np.testing.assert_allclose(q["x"], [2.25, 2.0 + 1.0 / 3.0], atol=1e-7)    # x_true + D^-1 b
    assert abs(q["residual_norm"] - 12.8198) < 1e-3
    z = O.neumann_solve(rp, ci, va, b, max_terms=20, series_tolerance=1e-8, start=O.START_ZERO,
                        residual=O.RESIDUAL_REFERENCE_SCALED)
    assert abs(z["residual_norm"] - 4.6015) < 1e-3


def test_neumann_error_order_and_failure():
    rp, ci, va = O.csr_from_triplets([0, 1], [0, 1], [2.0, 3.0], 2, 2)
    with pytest.raises(O.OracleError) as e:
        O.neumann_solve(rp, ci, va, [1.0, 2.0], cols=3)
    assert e.value.kind == "InvalidInput"                        # not square first (:147-152)
    with pytest.raises(O.OracleError) as e:
        O.neumann_solve(rp, ci, va, [1.0, 2.0, 3.0])
    assert e.value.kind == "DimensionMismatch"
    rp, ci, va = O.csr_from_triplets([0, 1], [0, 0], [2.0, 0.0], 2, 2)  # row 1 empty -> passes DD, fails diag
    with pytest.raises(O.OracleError) as e:
        O.neumann_solve(rp, ci, va, [1.0, 2.0])
    assert e.value.kind == "InvalidSparseMatrix"
    rp, ci, va = O.csr_from_triplets([0, 0, 1, 1], [0, 1, 0, 1], [4.0, 3.9, 3.9, 4.0], 2, 2)
    r = O.neumann_solve(rp, ci, va, [1.0, 1.0], max_iterations=10, max_terms=50, tolerance=1e-12, series_tolerance=1e-14)
    assert r["kind"] == "ConvergenceFailure" and r["iterations"] == 10


# ---- G5: push properties (tests/rust/push_tests.rs) ------------------------------------------------
def _four_node():
    # create_simple_graph, tests/rust/push_tests.rs:15-22
    return (np.array([0, 2, 4, 6, 7], np.uint32), np.array([1, 2, 0, 3, 0, 3, 1], np.uint32),
            np.array([0.5, 0.5, 0.8, 0.2, 0.6, 0.4, 1.0]))


def test_g5_acl_forward_push_fixture():
    rp, ci, w = _four_node()
    r = O.acl_push(rp, ci, w, [0], alpha=0.15, epsilon=1e-6)
    est, res = r["estimate"], r["residual"]
    assert (est >= 0).all() and (res >= 0).all()                                  # push_tests.rs:77-104
    assert abs(est.sum() + res.sum() - 1.0) < 1e-12                               # mass conservation :107-129
    # exact PPR: pi = alpha e_s^T (I - (1-alpha) P)^-1
    P = np.zeros((4, 4))
    for i in range(4):
        for k in range(rp[i], rp[i + 1]):
            P[i, ci[k]] = w[k]          # every row of the fixture already sums to 1
    pi = 0.15 * np.linalg.solve((np.eye(4) - 0.85 * P).T, np.eye(4)[0])
    np.testing.assert_allclose(est, pi, atol=2e-6)
    np.testing.assert_allclose(est, [0.431272, 0.276168, 0.183291, 0.109267], atol=2e-6)
    assert r["push_count"] == 200


def test_g5_monotone_and_edge_cases():
    rp, ci, w = _four_node()
    pushes = [O.acl_push(rp, ci, w, [0], epsilon=e)["push_count"] for e in (1e-2, 1e-4, 1e-6)]
    assert pushes[0] <= pushes[1] <= pushes[2]                                    # push_tests.rs:132-162
    r = O.acl_push(rp, ci, w, [0], alpha=0.99)
    assert r["estimate"][0] > 0.5                                                 # :520-537
    r = O.acl_push(rp, ci, w, [10])                                               # out of bounds source :433-495
    assert r["push_count"] == 0 and r["estimate"].sum() == 0.0
    # path graph, disconnected node keeps zero
    prp, pci, pw = O.csr_from_triplets([0, 1, 2], [1, 2, 3], [1.0, 1.0, 1.0], 5, 5)
    r = O.acl_push(prp, pci, pw, [0])
    assert r["estimate"][4] == 0.0 and r["estimate"][0] > 0
    b = O.acl_push(rp, ci, w, [3], backward=True)
    assert (b["estimate"] >= 0).all() and b["estimate"][3] > 0


def test_g5_backward_push_reference_tests():
    """tests/rust/push_tests.rs:255-302 — query_transition_probability, solve_multi_target, reachability_probabilities on the reference's
    own fixtures — plus solve_with_source's stop rule and empty results (backward_push.rs:238-293), which the reference leaves untested"""
    rp, ci, w = _four_node()
    q = lambda s, t: O.acl_push(rp, ci, w, [t], backward=True)["estimate"][s]
    assert 0.0 <= q(0, 3) <= 1.0 and q(0, 0) > 0.0                                # :256-267
    m = O.acl_push(rp, ci, w, [1, 3], backward=True)
    assert m["push_count"] > 0 and m["nodes_visited"] > 0 and m["estimate"][1] > 0 and m["estimate"][3] > 0      # :270-284
    prp, pci, pw = O.csr_from_triplets([0, 1, 2, 3], [1, 2, 3, 4], [1.0, 1.0, 1.0, 1.0], 5, 5)                   # create_path_graph(5), :49-58
    r = O.acl_push(prp, pci, pw, [4], backward=True)
    reach = O.acl_extrapolated_solution(0.15, r["estimate"], r["residual"])       # reachability_probabilities(4), :287-301
    assert reach[4] > reach[3] > reach[2] > reach[1]
    # The reference test's LAST inequality (reach[1] > reach[0] || reach[0] < 1e-6, :300) contradicts the code it tests: node 0 has no
    # in-edges, so backward_push_node keeps its mass on it (self loop, backward_push.rs:210-215) and every re-push adds alpha of it to the
    # estimate — 0.85^4 = 0.522 ends up on node 0.  The push files were never compiled, so the test never ran; the
    # restatement follows the CODE, and this pins what the code gives.
    assert abs(reach[0] - 0.85 ** 4) < 1e-5 and abs(reach[1] - 0.15 * 0.85 ** 3) < 1e-12
    full = O.acl_push(rp, ci, w, [3], backward=True)
    prec = 0.5 * full["estimate"][1]
    ws = O.acl_push(rp, ci, w, [3], backward=True, target=1, target_precision=prec)
    assert ws["push_count"] == 12 and full["push_count"] == 305                   # the loop ends at the first pop after the rule holds
    assert ws["estimate"][1] > prec and ws["residual"][1] < 0.1 * prec            # backward_push.rs:262-264
    for s, t in ((9, 3), (1, 9)):                                                 # :243-251
        e = O.acl_push(rp, ci, w, [t], backward=True, target=s, target_precision=0.1)
        assert e["push_count"] == 0 and e["estimate"].sum() == 0.0 and e["residual"].sum() == 0.0 and e["residual_norm"] == 0.0


# ---- G6: TS forward push (tests/mcp/mcp-tool-tests.js:27-52) ---------------------------------------
def _tridiag10():
    tr, tc, tv = [], [], []
    for i in range(10):
        for j, v in ((i - 1, -1.0), (i, 10.0), (i + 1, -1.0)):
            if 0 <= j < 10:
                tr.append(i), tc.append(j), tv.append(v)
    return O.csr_from_triplets(tr, tc, tv, 10, 10)


def test_g6_ts_forward_push():
    rp, ci, va = _tridiag10()
    b = np.zeros(10)
    b[0] = b[9] = 1.0
    r = O.ts_forward_push(rp, ci, va, b, 1e-3, 1000)
    assert r["status"] == 0 and r["converged"] and r["iterations"] == 12
    assert abs(r["residual"] - 5.2915e-4) < 1e-7
    np.testing.assert_allclose(b - O.spmv(rp, ci, va, r["x"]), r["r"], atol=1e-15)   # invariant r = b - A x


def test_sync_push_invariants_and_limit():
    rp, ci, va = _tridiag10()
    b = np.zeros(10)
    b[0] = b[9] = 1.0
    r = O.push_sync_solve(rp, ci, va, b, theta=1e-8, log_cap=4096)
    assert r["converged"]
    np.testing.assert_allclose(b - O.spmv(rp, ci, va, r["x"]), r["r"], atol=1e-14)
    np.testing.assert_allclose(r["x"], np.linalg.solve(_dense(rp, ci, va), b), atol=1e-8)
    log = r["frontier_log"]
    assert log[0] == 2 and log[1:3].tolist() == [0, 9]      # first frontier: the two loaded rows, ascending
    # theta larger than every scaled residual: nothing to push
    z = O.push_sync_solve(rp, ci, va, b, theta=1.0)
    assert z["rounds"] == 0 and z["converged"] and z["x"].sum() == 0.0


def _dense(rp, ci, va):
    n = len(rp) - 1
    A = np.zeros((n, n))
    for i in range(n):
        for k in range(rp[i], rp[i + 1]):
            A[i, ci[k]] += va[k]
    return A


# ---- G7: TS LCG stream (core/utils.ts:161-168) -----------------------------------------------------
def test_g7_lcg_streams():
    for seed in (1, 42, 12345):
        s = seed
        exp = []
        for _ in range(8):
            s = (s * 1664525 + 1013904223) % 2 ** 32
            exp.append(s / 2 ** 32)
        assert O.ts_lcg(seed, 8).tolist() == exp
    assert O.ts_lcg(1, 1)[0] == 1015568748 / 2 ** 32


def test_random_walk_estimate_runs_and_is_seeded():
    rp, ci, va = _tridiag10()
    b = np.ones(10)
    m1 = O.ts_random_walk_estimate(rp, ci, va, b, 0, 0.1, 42)
    m2 = O.ts_random_walk_estimate(rp, ci, va, b, 0, 0.1, 42)
    assert m1 == m2 and m1[2] == 100


def test_g9_pagerank_system_matches_the_reference_power_iteration():
    """G9: the PageRank system of computePageRank (solver.ts:664-722), assembled by generators.pagerank_system and solved by the
    oracle's thresholded push, against the fixed point of the reference's runnable power iteration
    (scripts/pagerank/sublinear_pagerank.py:145-166; tests/golden/make_golden_pagerank.py): dangling nodes, self loops, weights, a hub"""
    import scipy.sparse as sp
    from pathlib import Path
    from sublinear_time_solver_amd import generators as G
    z = np.load(Path(__file__).resolve().parent / "golden" / "reference_pagerank.npz")
    for key in (str(c) for c in z["__cases"]):
        n, d = int(z[f"{key}__n"][0]), float(z[f"{key}__damping"][0])
        A = sp.csr_matrix((z[f"{key}__vals"], (z[f"{key}__rows"].astype(np.int64), z[f"{key}__cols"].astype(np.int64))), shape=(n, n))
        rp, ci, va, b = G.pagerank_system(n, A.indptr.astype(np.uint32), A.indices.astype(np.uint32), A.data, d)
        r = O.push_sync_solve(rp, ci, va, b, theta=1e-18)
        ref = z[f"{key}__pagerank"]
        assert r["converged"] and np.abs(r["x"] - ref).max() <= 1e-13, (key, np.abs(r["x"] - ref).max())


def test_oracle_is_clean_under_asan_and_ubsan():
    """`make -C oracle asan`: every oracle entry point on empty / tiny / ragged / duplicate-laden / hub systems under
    AddressSanitizer + UndefinedBehaviorSanitizer (oracle/asan_check.c).  The oracle is the checker of the GPU parity tests."""
    import subprocess
    from pathlib import Path
    r = subprocess.run(["make", "-C", str(Path(__file__).resolve().parent.parent / "oracle"), "asan"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "asan_check ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


# ---- G12: the reference's own TypeScript forward push, executed (tests/golden/make_golden_ts_push.py) -----------------------------------
def test_g12_ts_forward_push_equals_the_executed_reference():
    """solveForwardPush (src/core/solver.ts:437-522) as the reference's own code ran it on six seeded systems: iterations, solution bits,
    the residual norm bit for bit (norm2's sequential reduce), and the CONVERGENCE_FAILED exit with the residual it reports"""
    from pathlib import Path
    g = np.load(Path(__file__).resolve().parent / "golden" / "reference_ts_push.npz")
    kinds = set()
    for k in map(str, g["names"]):
        n, maxit, conv, its = (int(v) for v in g[k + "/params"])
        eps, res = (float(v) for v in g[k + "/epsilon_residual"])
        rp, ci, va = O.csr_from_triplets(g[k + "/rows"], g[k + "/cols"], g[k + "/values"], n, n)
        r = O.ts_forward_push(rp, ci, va, g[k + "/b"], eps, maxit)
        assert r["iterations"] == its and r["converged"] == bool(conv) and r["residual"] == res, (k, r["iterations"], its, r["residual"], res)
        assert r["status"] == (0 if conv else 3), k
        if conv:
            assert (r["x"].view(np.uint64) == g[k + "/solution"].view(np.uint64)).all(), k
        kinds.add(bool(conv))
    assert kinds == {True, False} and len(g["names"]) == 6


# ---- G13: the reference's own TypeScript computePageRank, executed (tests/golden/make_golden_ts_pagerank.py) ----------------------------
def golden_ts_pagerank_cases():
    from pathlib import Path
    g = np.load(Path(__file__).resolve().parent / "golden" / "reference_ts_pagerank.npz")
    for k in map(str, g["names"]):
        n, maxit, its, pers = (int(v) for v in g[k + "/params"])
        d, eps, res = (float(v) for v in g[k + "/damping_epsilon_residual"])
        yield k, g, dict(n=n, maxit=maxit, its=its, personalized=bool(pers), damping=d, eps=eps, residual=res)


def test_g13_pagerank_system_and_solve_equal_the_executed_reference():
    """computePageRank (src/core/solver.ts:664-722, method forward-push) as the reference's own code ran it: the assembled system matrix —
    out-degrees as left-to-right row sums, [i == j] - damping * (adj[j][i] / out_j), dangling columns untouched, self loops on the
    diagonal — entry for entry and bit for bit, the right-hand side, and the solution / iteration count / residual of its own push"""
    from sublinear_time_solver_amd import generators as G
    n_cases = 0
    for k, g, c in golden_ts_pagerank_cases():
        n = c["n"]
        arp, aci, ava = G.adjacency_csr_first_match(g[k + "/adj_rows"], g[k + "/adj_cols"], g[k + "/adj_values"], n)
        rp, ci, va, b = G.pagerank_system(n, arp, aci, ava, c["damping"])
        S = g[k + "/system"]
        rr, cc = np.nonzero(S)
        srp, sci, sva = O.csr_from_triplets(rr, cc, S[rr, cc], n, n)
        assert (srp == rp).all() and (sci == ci).all() and (sva.view(np.uint64) == va.view(np.uint64)).all(), k
        if not c["personalized"]:
            assert (b.view(np.uint64) == g[k + "/rhs"].view(np.uint64)).all(), k
        r = O.ts_forward_push(rp, ci, va, g[k + "/rhs"], c["eps"], c["maxit"])
        assert r["iterations"] == c["its"] and r["residual"] == c["residual"] and (r["x"].view(np.uint64) == g[k + "/solution"].view(np.uint64)).all(), k
        n_cases += 1
    assert n_cases == 3


def test_adjacency_is_read_the_way_the_reference_reads_a_coo_matrix():
    """MatrixOperations.getEntry returns the FIRST stored match (matrix.ts:105-112): later duplicates never count, a stored 0 hides them"""
    from sublinear_time_solver_amd import generators as G
    r, c, v = [0, 0, 1, 0, 2, 2], [1, 1, 0, 2, 0, 0], [2.0, 5.0, 1.0, 3.0, 0.0, 7.0]
    rp, ci, va = G.adjacency_csr_first_match(r, c, v, 3)
    assert rp.tolist() == [0, 2, 3, 3] and ci.tolist() == [1, 2, 0] and va.tolist() == [2.0, 3.0, 1.0]


# ---- G15: the reference's own `generate -t diagonally-dominant`, executed with its seeded generator (tests/golden/make_golden_ts_generate.py) ----
def test_g15_config0_generator_equals_the_executed_reference():
    """generators.gen1000_dense against MatrixTools.generateDiagonallyDominantMatrix (src/mcp/tools/matrix.ts:297-322) run by node with
    Math.random = createSeededRandom(seed): every entry at size 150, and BASELINE config 0 itself (size 1000, strength 2, seed 42) by entry
    count, exact sums and the SHA-256 of the dense table"""
    import hashlib
    import math
    from pathlib import Path
    from sublinear_time_solver_amd import generators as G
    g = np.load(Path(__file__).resolve().parent / "golden" / "reference_ts_generate.npz")
    assert [str(k) for k in g["names"]][-1] == "dd1000_strength2_seed42"
    for k in map(str, g["names"]):
        size, seed, nnz = (int(v) for v in g[k + "/params"])
        rp, ci, va, b = G.gen1000_dense(size, float(g[k + "/strength"][0]), seed)
        a = np.zeros((size, size))
        a[np.repeat(np.arange(size), np.diff(rp.astype(np.int64))), ci] = va
        assert int((a != 0).sum()) == nnz and hashlib.sha256(a.tobytes()).digest() == bytes(g[k + "/sha256"]), k
        assert (math.fsum(np.diag(a).tolist()), math.fsum(a.ravel().tolist())) == tuple(g[k + "/sums"]), k
        if k + "/values" in g:
            rr, cc = np.nonzero(a)
            assert (rr == g[k + "/rows"]).all() and (cc == g[k + "/cols"]).all() and (a[rr, cc].view(np.uint64) == g[k + "/values"].view(np.uint64)).all(), k
