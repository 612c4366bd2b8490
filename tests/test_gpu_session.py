"""Query sessions (sl_query_session_*): single-entry queries whose cost follows the touched rows.  Checked against
the CPU restatement of the push on A^T from e_row (same rounds, same pushes, estimate = y.b to rounding), against
the one-shot entry points, and for state hygiene: a session answers the same query with the same bits however many
other queries ran in between, including ones that flood the graph and ones that stop at the round limit."""
import numpy as np
import pytest

import sublinear_time_solver_amd as S
from sublinear_time_solver_amd import generators as G
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _oracle_query(rp, ci, va, n, b, row, theta, max_rounds=100_000):
    trp, tci, tva = O.csr_transpose(rp, ci, va, n)
    e = np.zeros(n)
    e[row] = 1.0
    o = O.push_sync_solve(trp, tci, tva, e, theta=theta, max_rounds=max_rounds)
    return o, float(np.dot(o["x"], b)), float(np.abs(o["r"]).sum())


@pytest.mark.parametrize("n,k,w", [(20_000, 8, 0), (30_011, 13, 700)])
def test_session_queries_match_cpu_push_and_one_shot(gpu, n, k, w):
    rp, ci, va, b = G.sdd_rows(n, k, seed=5, half_bandwidth=w)
    m = S.SparseMatrix.from_csr(rp, ci, va, n, n, with_transpose=True)
    with S.QuerySession(m, b) as q:
        first = {}
        for rnd in range(2):                                  # second pass: same bits after other queries ran
            for row, theta in [(0, 1e-4), (n // 3, 1e-6), (n - 1, 1e-9), (17, 1e-2), (n // 2, 10.0)]:
                e = q.estimate(row, theta=theta)
                if rnd == 0:
                    o, est, l1 = _oracle_query(rp, ci, va, n, b, row, theta)
                    assert e.rounds == o["rounds"] and e.pushes == o["pushes"] and bool(e.converged)
                    assert abs(e.estimate - est) <= 1e-14 * max(1.0, abs(est)) and abs(e.residual_l1 - l1) <= 1e-14 * max(1.0, l1)
                    one = S.estimate_entry(m, b, row, theta=theta)
                    assert (one.rounds, one.pushes, one.rows_touched) == (e.rounds, e.pushes, e.rows_touched)
                    assert abs(one.estimate - e.estimate) <= 1e-15 * max(1.0, abs(est))
                    first[(row, theta)] = (e.estimate, e.residual_l1, e.rounds, e.pushes, e.rows_touched)
                else:
                    assert first[(row, theta)] == (e.estimate, e.residual_l1, e.rounds, e.pushes, e.rows_touched)
        # theta above |1/a_ii|: nothing is pushed, the residual is the seed itself
        e = q.estimate(5, theta=10.0)
        assert e.rounds == 0 and e.pushes == 0 and e.estimate == 0.0 and e.residual_l1 == 1.0 and bool(e.converged)


def test_session_survives_flooding_and_round_limit(gpu):
    n = 4096
    rp, ci, va, b = G.sdd_rows(n, 8, seed=2, half_bandwidth=0)
    m = S.SparseMatrix.from_csr(rp, ci, va, n, n, with_transpose=True)
    x = S.NeumannSolver().solve(m, b, S.SolverOptions(tolerance=1e-13)).solution
    with S.QuerySession(m, b) as q:
        ref = q.estimate(100, theta=1e-3)
        flood = q.estimate(7, theta=1e-14)                     # floods: dense rounds, whole-vector reset afterwards
        assert bool(flood.converged) and abs(flood.estimate - x[7]) <= 1e-11
        again = q.estimate(100, theta=1e-3)
        assert (again.estimate, again.residual_l1, again.rounds, again.pushes) == (ref.estimate, ref.residual_l1, ref.rounds, ref.pushes)
        cut = q.estimate(9, theta=1e-9, max_rounds=3)          # stops with a live frontier: leftovers must be cleaned up
        o, est, l1 = _oracle_query(rp, ci, va, n, b, 9, 1e-9, max_rounds=3)
        assert cut.rounds == 3 and not bool(cut.converged) and cut.pushes == o["pushes"]
        assert abs(cut.estimate - est) <= 1e-14 and abs(cut.residual_l1 - l1) <= 1e-13
        again = q.estimate(100, theta=1e-3)
        assert (again.estimate, again.residual_l1, again.rounds, again.pushes) == (ref.estimate, ref.residual_l1, ref.rounds, ref.pushes)
    mt = m.transpose(with_transpose=True)
    with S.QuerySession(mt, b, matrix_is_transpose=True) as qt:  # the caller already holds A^T
        e = qt.estimate(100, theta=1e-3)
        assert (e.rounds, e.pushes) == (ref.rounds, ref.pushes) and abs(e.estimate - ref.estimate) <= 1e-15


def test_session_sums_are_exact_to_80_bits_and_order_free(gpu):
    """estimate = sum_i y_i b_i over the touched rows with a right-hand side spanning 40 decades and alternating signs: the binned
    sum must agree with the correctly rounded sum (math.fsum of the CPU terms) far below fp64 rounding of a naive sum, and must not
    depend on the order in which the rounds appended the touched rows (same bits on every repetition and from the one-shot call)"""
    import math
    n, k, w = 25_000, 11, 300
    rp, ci, va, _ = G.sdd_rows(n, k, seed=8, half_bandwidth=w)
    i = np.arange(n)
    b = np.where(i % 2 == 0, 1.0, -1.0) * 10.0 ** ((i * 7) % 41 - 20)
    m = S.SparseMatrix.from_csr(rp, ci, va, n, n, with_transpose=True)
    row, theta = n // 2, 1e-9
    o, _, _ = _oracle_query(rp, ci, va, n, b, row, theta)
    terms = o["x"] * b
    exact = math.fsum(terms.tolist())
    l1_exact = math.fsum(np.abs(o["r"]).tolist())
    top = float(np.abs(terms).max())
    with S.QuerySession(m, b) as q:
        got = [q.estimate(row, theta=theta) for _ in range(3)]
    assert got[0].rounds == o["rounds"] and np.count_nonzero(terms) > 1000
    assert abs(got[0].estimate - exact) <= 2.0 ** -52 * abs(exact) + np.count_nonzero(terms) * 2.0 ** -80 * top
    assert abs(got[0].residual_l1 - l1_exact) <= 2.0 ** -52 * l1_exact
    assert all((g.estimate, g.residual_l1) == (got[0].estimate, got[0].residual_l1) for g in got)
    one = S.estimate_entry(m, b, row, theta=theta)
    assert (one.estimate, one.residual_l1) == (got[0].estimate, got[0].residual_l1)


def test_batch_of_queries_on_lanes_equals_one_at_a_time(gpu):
    """sl_query_session_estimate_batch: independent queries on lanes (own state, stream and host thread each) — every result equals the
    one-at-a-time answer bit for bit, whatever lane ran it and whatever ran beside it; the session stays usable afterwards"""
    import sublinear_time_solver_amd as S
    from sublinear_time_solver_amd import generators as G
    n = 200_000
    rp, ci, va = G.pagerank_graph(n, 5)
    rp2, ci2, va2, b = G.pagerank_system(n, rp, ci, va, 0.85)
    m = S.SparseMatrix.from_csr(rp2, ci2, va2, n, n, with_transpose=True)
    rows = [0, 17, n - 1, n // 2, 12345, 99_999, 3, n - 2, 54_321, 1] * 3                 # repeats included
    with S.QuerySession(m, b) as sess:
        one = [sess.estimate(r, theta=1e-7) for r in rows]
        for lanes in (1, 3, 8):
            got = sess.estimate_batch(rows, theta=1e-7, lanes=lanes)
            assert len(got) == len(rows)
            for a, g in zip(one, got):
                assert np.float64(a.estimate).view(np.uint64) == np.float64(g.estimate).view(np.uint64)
                assert np.float64(a.residual_l1).view(np.uint64) == np.float64(g.residual_l1).view(np.uint64)
                assert (a.rounds, a.pushes, a.rows_touched, a.converged) == (g.rounds, g.pushes, g.rows_touched, g.converged)
        again = sess.estimate(rows[2], theta=1e-7)                                        # lane 0's state is all-zero again
        assert np.float64(again.estimate).view(np.uint64) == np.float64(one[2].estimate).view(np.uint64)
        with pytest.raises(S.SolverError):
            sess.estimate_batch([0, n + 7], theta=1e-7)


def test_small_rounds_in_one_workgroup_give_the_same_answers(gpu):
    """SL_PUSH_SMALL=1: rounds with small frontiers run back to back in ONE workgroup (sl_small_rounds_kernel) instead of four launches
    each.  Same phases, same records, same ordered sums: every query answers with the same bits — estimate, residual, rounds, pushes, rows
    touched — as without it, for queries that stay small, queries whose middle rounds need the launch train, queries that flood, queries cut
    off by the round limit, on a graph with hub columns (long-column pieces, heavy rows) and on S-DD; and push solves give the same x / r."""
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
out = []
def bits(v): return int(np.float64(v).view(np.uint64))
n = 60_000
rp, ci, va, b = G.sdd_rows(n, 11, seed=9, half_bandwidth=0)
m = S.SparseMatrix.from_csr(rp, ci, va, n, n, with_transpose=True)
with S.QuerySession(m, b) as q:
    for row, theta, mr in [(0, 1e-3, 100000), (n // 3, 1e-5, 100000), (n - 1, 1e-7, 100000), (17, 1e-9, 100000), (5, 1e-6, 3), (n // 2, 1e-12, 100000), (0, 1e-3, 100000)]:
        e = q.estimate(row, theta=theta, max_rounds=mr)
        out.append([bits(e.estimate), bits(e.residual_l1), int(e.rounds), int(e.pushes), int(e.rows_touched), int(bool(e.converged))])
    for e in q.estimate_batch([3, 99, 4242, n - 7], theta=1e-5, lanes=3):
        out.append([bits(e.estimate), bits(e.residual_l1), int(e.rounds), int(e.pushes), int(e.rows_touched)])
# a power-law graph: hub columns (pieces), heavy rows
adj = G.pagerank_graph(20_000, 3)
prp, pci, pva, pb = G.pagerank_system(20_000, *adj, damping=0.85)
pm = S.SparseMatrix.from_csr(prp, pci, pva, 20_000, 20_000, with_transpose=True)
with S.QuerySession(pm, pb) as q:
    for row, theta in [(0, 1e-4), (7, 1e-6), (19_999, 1e-5), (123, 1e-8)]:
        e = q.estimate(row, theta=theta)
        out.append([bits(e.estimate), bits(e.residual_l1), int(e.rounds), int(e.pushes), int(e.rows_touched)])
bs = pb * (np.arange(20_000) % 97 == 0)
p = S.PushSolver(theta=1e-7).solve(pm, bs)
out.append([int(p["rounds"]), int(p["pushes"]), int(p["solution"].view(np.uint64).sum() % (1 << 61)), int(p["residual"].view(np.uint64).sum() % (1 << 61))])
print("RESULT " + json.dumps(out))
"""
    res = {}
    for mode in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", prog], cwd=root, capture_output=True, text=True, timeout=900, env=dict(os.environ, SL_PUSH_SMALL=mode))
        assert r.returncode == 0, r.stderr[-3000:]
        res[mode] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    assert res["0"] == res["1"], [(a, b) for a, b in zip(res["0"], res["1"]) if a != b][:3]
    assert any(row[2] >= 6 for row in res["0"][:7])             # several rounds did run


def test_wide_batch_answers_equal_single_queries(gpu):
    """SL_QUERY_WIDE=W: W queries share ONE launch train (grid.y = slot).  Every answer — estimate, residual, rounds, pushes, rows touched,
    converged — equals the one-at-a-time answer bit for bit: local queries, a seed below its threshold (no round at all), queries the batch
    does not finish (they flood, or need more rounds than a batch holds: the slot is cleaned and the ordinary path answers), a round limit
    inside the batch, a count that is not a multiple of W, the same row twice; with and without the one-workgroup small rounds."""
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
def bits(v): return int(np.float64(v).view(np.uint64))
def rec(e): return [bits(e.estimate), bits(e.residual_l1), int(e.rounds), int(e.pushes), int(e.rows_touched), int(bool(e.converged))]
out = {}
n = 50_000
rp, ci, va, b = G.sdd_rows(n, 9, seed=12, half_bandwidth=0)
m = S.SparseMatrix.from_csr(rp, ci, va, n, n, with_transpose=True)
rows = [0, 17, n - 1, 4242, 17, 31337, 9, 25_000, 40_001, 3, 12_345]
with S.QuerySession(m, b) as q:
    for theta, mr in [(1e-4, 100000), (1e-6, 100000), (10.0, 100000), (1e-7, 3), (1e-13, 100000)]:
        out[f"batch {theta} {mr}"] = [rec(e) for e in q.estimate_batch(rows, theta=theta, max_rounds=mr)]
        out[f"single {theta} {mr}"] = [rec(q.estimate(r, theta=theta, max_rounds=mr)) for r in rows]
adj = G.pagerank_graph(20_000, 5)
prp, pci, pva, pb = G.pagerank_system(20_000, *adj, damping=0.85)
pm = S.SparseMatrix.from_csr(prp, pci, pva, 20_000, 20_000, with_transpose=True)
prow = [0, 5, 19_999, 777, 10_000, 64, 1]
with S.QuerySession(pm, pb) as q:
    out["pr batch"] = [rec(e) for e in q.estimate_batch(prow, theta=1e-5)]
    out["pr single"] = [rec(q.estimate(r, theta=1e-5)) for r in prow]
print("RESULT " + json.dumps(out))
"""
    for extra in ({}, {"SL_PUSH_SMALL": "1"}):
        r = subprocess.run([sys.executable, "-c", prog], cwd=root, capture_output=True, text=True, timeout=900, env=dict(os.environ, SL_QUERY_WIDE="4", **extra))
        assert r.returncode == 0, r.stderr[-3000:]
        res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
        for key in res:
            if key.startswith("batch") or key == "pr batch":
                other = key.replace("batch", "single")
                # device_time aside, the records must be identical
                assert res[key] == res[other], (extra, key, [(a, b) for a, b in zip(res[key], res[other]) if a != b][:2])
        assert all(rr[2] == 0 and rr[0] == 0 for rr in res["batch 10.0 100000"])          # seeds below the threshold: no round, estimate 0
        assert all(rr[2] == 3 and rr[5] == 0 for rr in res["batch 1e-07 3"])               # cut off by the round limit inside the batch
        assert any(rr[2] > 12 for rr in res["batch 1e-13 100000"])                         # more rounds than a batch holds: answered by the ordinary path
