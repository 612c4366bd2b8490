"""Monte-Carlo estimateEntry branch and the `random-walk` method of solve(): GPU walks vs the oracle restatement of the
same rule — the reference's ONE LCG stream cut into a block of 2048 draws per walk — per-walk values bit-identical, mean / variance to
1e-12; and against the reference as written (the stream walked serially) within Monte-Carlo error."""
import ctypes as C

import numpy as np
import pytest

import sublinear_time_solver_amd as S
from sublinear_time_solver_amd import _lib as L
from sublinear_time_solver_amd import generators as G
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _walk(m, b, row, n_samples, seed, eps=0.1, stream=L.SL_WALK_STREAM_BLOCKS):
    lib = L.load()
    vals = np.zeros(max(n_samples, 1))
    res = L.WalkResult()
    L.check(lib.sl_estimate_entry_random_walk(m._h, L.ptr(np.ascontiguousarray(b, dtype=np.float64)), 0, row, eps, seed, stream, n_samples,
                                              L.ptr(vals) if n_samples else None, C.byref(res)))
    return vals, res


@pytest.mark.parametrize("n,k,w,seed", [(2000, 8, 0, 42), (5000, 16, 50, 1), (300, 5, 0, 12345)])
def test_walk_values_bitwise_vs_oracle(gpu, n, k, w, seed):
    rp, ci, va, _ = G.sdd_rows(n, k, seed=3, half_bandwidth=w)
    b = np.random.default_rng(n).standard_normal(n) * 2.0        # (the generator's own b has b_i / a_ii = 0.1 for every i: every walk would return 0.1)
    m = S.SparseMatrix.from_csr(rp, ci, va, n, n, keep_csr=True)
    N = 20000
    for row in (0, n // 2):
        gv, res = _walk(m, b, row, N, seed)
        ov, om, ovar = O.ts_random_walk_streams(rp, ci, va, b, row, N, seed)
        assert (gv.view(np.uint64) == ov.view(np.uint64)).all(), "per-walk values must be bit-identical"
        assert np.unique(ov).size > 50
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


def test_random_walk_solve_against_the_oracle(gpu):
    """solveRandomWalk (core/solver.ts:278-357), the `random-walk` METHOD of solve(): every coordinate from its own walks.  Against the
    oracle's per-walk-stream restatement: means / variances / residual to 1e-12 (tree-reduced sums), the verdict and the walk count equal;
    against the reference AS WRITTEN (one shared stream, oracle per_walk_streams = 0): within Monte-Carlo error, coordinate by coordinate."""
    n, eps, seed = 60, 0.05, 9
    rp, ci, va, _ = G.sdd_rows(n, 6, seed=5)
    b = np.random.default_rng(1).standard_normal(n) * 3.0            # (the generator's own b has b_i / a_ii = 0.1 for every i: all walks equal)
    m = S.SparseMatrix.from_csr(rp, ci, va, n, n, keep_csr=True)
    r = S.random_walk_solve(m, b, eps, seed)
    o = O.ts_random_walk_solve(rp, ci, va, b, eps, seed, per_walk_streams=True)
    W = 400
    assert r["num_walks"] == W == max(100, int(np.ceil(1 / eps ** 2))) and r["iterations"] == n
    assert np.abs(r["solution"] - o["x"]).max() <= 1e-12 * np.abs(o["x"]).max()
    assert np.abs(r["variances"] - o["variances"]).max() <= 1e-12 * o["variances"].max() and o["variances"].min() > 0
    assert abs(r["residual"] - o["residual"]) <= 1e-10 * o["residual"] and abs(r["total_variance"] - o["total_variance"]) <= 1e-12 * o["total_variance"]
    assert r["converged"] == o["converged"]
    # each coordinate IS the single-entry estimate (sl_estimate_entry_random_walk) seeded where that coordinate's walks begin in the stream
    for i in (0, 17, n - 1):
        at = O.ts_lcg_jump(seed, i * W * 2048)
        gv, res = _walk(m, b, i, W, at)
        assert abs(res.estimate - r["solution"][i]) <= 1e-12 * abs(res.estimate) and (gv.view(np.uint64) == O.ts_random_walk_streams(rp, ci, va, b, i, W, at)[0].view(np.uint64)).all()
    # the reference's own single stream: a different sample of the same estimator
    s = O.ts_random_walk_solve(rp, ci, va, b, eps, seed, per_walk_streams=False)
    z = np.abs(r["solution"] - s["x"]) / np.sqrt((r["variances"] + s["variances"]) / W)
    assert z.max() < 6.0 and np.mean(z < 2.0) > 0.8, (z.max(), np.mean(z < 2.0))
    # reproducible, and the seed matters
    again = S.random_walk_solve(m, b, eps, seed)
    assert (again["solution"].view(np.uint64) == r["solution"].view(np.uint64)).all()
    assert (S.random_walk_solve(m, b, eps, seed + 1)["solution"] != r["solution"]).any()


def test_random_walk_method_of_the_ts_surface(gpu):
    """SublinearSolver({method: 'random-walk'}).solve: converges where the walks are exact (a diagonal system: every walk ends at its
    start with b_i / a_ii), throws CONVERGENCE_FAILED where the residual misses epsilon (solver.ts:335-341), as the reference does"""
    d = [4.0, -5.0, 8.0]
    A = {"rows": 3, "cols": 3, "format": "dense", "data": [[d[0], 0, 0], [0, d[1], 0], [0, 0, d[2]]]}
    seen = []
    out = S.SublinearSolver(method="random-walk", epsilon=0.1, seed=3).solve(A, [1.0, 2.0, 3.0], progress_callback=seen.append)
    assert len(seen) == 1 and seen[0]["iteration"] == 3 and seen[0]["residual"] == out["residual"]
    assert out["converged"] and out["method"] == "random-walk" and out["iterations"] == 3 and out["residual"] < 1e-15
    assert np.allclose(out["solution"], [0.25, -0.4, 0.375], rtol=0, atol=1e-16)
    B = {"rows": 2, "cols": 2, "format": "coo", "values": [4.0, 1.0, 1.0, 3.0], "rowIndices": [0, 0, 1, 1], "colIndices": [0, 1, 0, 1]}
    with pytest.raises(S.SolverError) as e:
        S.SublinearSolver(method="random-walk", epsilon=0.05, seed=3).solve(B, [5.0, 4.0])
    assert e.value.status == 3 and "Random walk sampling failed" in str(e.value)
    with pytest.raises(S.SolverError) as e:                             # createTransitionMatrix: "Zero diagonal at position 1" (solver.ts:368-371)
        S.random_walk_solve(S.SparseMatrix.from_triplets([(0, 0, 1.0), (0, 1, 1.0), (1, 0, 1.0)], 2, 2, keep_csr=True), [1.0, 1.0], 0.1, 1)
    assert e.value.status == 2 and "Zero diagonal at position 1" in str(e.value)


# ---- SL_WALK_STREAM_SERIAL: the reference AS WRITTEN (one stream walked serially) — bit for bit ------------------------------------------
def test_serial_stream_equals_the_executed_reference(gpu):
    """tests/golden/reference_walk.npz holds what the reference's own TypeScript walk code printed for these inputs
    (tests/golden/make_golden_walk.py): every per-walk value, estimate and variance of estimateEntry, every x_i / variance_i /
    totalVariance of solveRandomWalk must come back from the device with the same bits under stream = serial"""
    from tests.test_oracle_walk import golden_walk_cases
    kinds = set()
    for k, g, c in golden_walk_cases():
        m = S.SparseMatrix.from_csr(c["rp"], c["ci"], c["va"], c["n"], c["n"], keep_csr=True)
        if k + "/estimates" in g:
            want = g[k + "/estimates"]
            vals, res = _walk(m, c["b"], c["row"], 0, c["seed"], eps=c["eps"], stream=L.SL_WALK_STREAM_SERIAL)      # the count from epsilon, as the reference derives it
            assert res.num_samples == want.size, k
            vals, res = _walk(m, c["b"], c["row"], want.size, c["seed"], eps=c["eps"], stream=L.SL_WALK_STREAM_SERIAL)
            assert (vals.view(np.uint64) == want.view(np.uint64)).all(), k
            assert (res.estimate, res.variance) == tuple(g[k + "/mean_variance"]), k
            kinds.add("estimate")
        else:
            r = S.random_walk_solve(m, c["b"], c["eps"], c["seed"], stream="reference")
            assert (r["solution"].view(np.uint64) == g[k + "/solution"].view(np.uint64)).all(), k
            assert (r["variances"].view(np.uint64) == g[k + "/variances"].view(np.uint64)).all(), k
            tv, res = g[k + "/total_variance_residual"]
            assert r["total_variance"] == tv, k
            assert abs(r["residual"] - res) <= 1e-12 * res, k                       # (the norm of the residual is a tree reduction on the device)
            assert not r["converged"]                                               # these systems miss epsilon: the reference throws there
            kinds.add("solve")
    assert kinds == {"estimate", "solve"}


def test_serial_stream_against_the_oracle_on_larger_systems(gpu):
    n, seed = 3000, 77
    rp, ci, va, _ = G.sdd_rows(n, 8, seed=3, half_bandwidth=60)
    b = np.random.default_rng(4).standard_normal(n)
    m = S.SparseMatrix.from_csr(rp, ci, va, n, n, keep_csr=True)
    for row, N in ((0, 2500), (n - 1, 400)):
        gv, res = _walk(m, b, row, N, seed, stream=L.SL_WALK_STREAM_SERIAL)
        ov, om, ovar = O.ts_random_walk_serial(rp, ci, va, b, row, N, seed)
        assert (gv.view(np.uint64) == ov.view(np.uint64)).all() and (res.estimate, res.variance) == (om, ovar)
        bv, _ = _walk(m, b, row, N, seed)                                           # the block form: walk 0 is the same walk, the rest differ
        assert bv[0] == gv[0] and (bv != gv).any()
    # the TS surface takes {stream: 'reference'}
    tri = {"rows": 3, "cols": 3, "format": "coo", "values": [4.0, -1.0, -1.0, 5.0, -2.0, -1.0, 6.0], "rowIndices": [0, 0, 1, 1, 1, 2, 2],
           "colIndices": [0, 1, 0, 1, 2, 1, 2]}
    trp, tci, tva = O.csr_from_triplets(tri["rowIndices"], tri["colIndices"], tri["values"], 3, 3)
    out = S.SublinearSolver(method="random-walk", epsilon=0.1, seed=11, stream="reference").estimate_entry(tri, [1.0, 2.0, 3.0], row=1, method="random-walk")
    om, ovar, ns = O.ts_random_walk_estimate(trp, tci, tva, [1.0, 2.0, 3.0], 1, 0.1, 11)
    assert (out["estimate"], out["variance"], out["numSamples"]) == (om, ovar, ns)
    with pytest.raises(S.SolverError):
        S.SublinearSolver(method="random-walk", stream="nonsense")


def test_serial_stream_pipeline_equals_the_one_lane_kernel(gpu, monkeypatch):
    """the default SERIAL form simulates a walk from EVERY stream position and then follows the reference's chain through them
    (sl_walk.hip, "the serial stream, data-parallel"); SL_WALK_SERIAL_PLAIN=1 is the reference as written on one lane.  Same bits, and the
    oracle's — on walks that run into the 1000-step cap (2000 draws: the longest jump a chunk's table must carry), on windows of one chunk
    (every window boundary is a chunk boundary the chain crosses), on windows that yield fewer walks than asked for, across coordinates"""
    n = 48
    rows, cols, vals = [], [], []
    for i in range(n):                                                  # a chain; a walk stops with probability 1 / a_ii per step (solver.ts:398-401): from rows
        d = 3000.0 if i < 16 else 2.0 + 0.5 * (1 + i % 5)               # 0..15 it runs for hundreds of steps, many into the cap; the other rows absorb quickly
        for j, v in ((i - 1, -1.0), (i, d), (i + 1, -1.0)):
            if 0 <= j < n:
                rows.append(i), cols.append(j), vals.append(v)
    rp, ci, va = O.csr_from_triplets(rows, cols, vals, n, n)
    b = np.random.default_rng(12).standard_normal(n)
    m = S.SparseMatrix.from_csr(rp, ci, va, n, n, keep_csr=True)
    for window, row, N, seed in (("", 3, 300, 5), ("4096", 3, 150, 6), ("4096", 30, 3000, 7), ("8192", 17, 900, 8), ("", 40, 5000, 9)):
        ov, om, ovar = O.ts_random_walk_serial(rp, ci, va, b, row, N, seed)
        assert row != 3 or (ov == 0.0).any()                            # (a walk cut off at 1000 steps contributes 0: such walks are in the sample)
        got = {}
        for plain in ("0", "1"):
            monkeypatch.setenv("SL_WALK_SERIAL_PLAIN", plain)
            monkeypatch.setenv("SL_WALK_SPEC_WINDOW", window) if window else monkeypatch.delenv("SL_WALK_SPEC_WINDOW", raising=False)
            gv, res = _walk(m, b, row, N, seed, stream=L.SL_WALK_STREAM_SERIAL)
            got[plain] = (gv.copy(), res.estimate, res.variance)
            assert (gv.view(np.uint64) == ov.view(np.uint64)).all() and (res.estimate, res.variance) == (om, ovar), (window, row, plain)
        assert (got["0"][0].view(np.uint64) == got["1"][0].view(np.uint64)).all() and got["0"][1:] == got["1"][1:]
    sol = {}
    for plain, window in (("1", ""), ("0", ""), ("0", "4096")):         # solveRandomWalk: one stream through all coordinates
        monkeypatch.setenv("SL_WALK_SERIAL_PLAIN", plain)
        monkeypatch.setenv("SL_WALK_SPEC_WINDOW", window) if window else monkeypatch.delenv("SL_WALK_SPEC_WINDOW", raising=False)
        r = S.random_walk_solve(m, b, 0.2, 21, stream="reference")
        sol[(plain, window)] = np.concatenate([r["solution"], r["variances"], [r["total_variance"]]]).view(np.uint64)
    assert (sol[("1", "")] == sol[("0", "")]).all() and (sol[("1", "")] == sol[("0", "4096")]).all()
    o = O.ts_random_walk_solve(rp, ci, va, b, 0.2, 21)                  # (the reference's one shared stream)
    assert (sol[("0", "")] == np.concatenate([o["x"], o["variances"], [o["total_variance"]]]).view(np.uint64)).all()


def test_row_table_changes_no_bit(gpu, monkeypatch):
    """a call that expects its walks to visit at least as many rows as the matrix has first tabulates every row's diagonal and the sum of
    its transition weights (sl_walk_table_kernel) — two of walk_one's three passes over a row, done once per row instead of once per
    visit.  Same operations in the same order: SL_WALK_TABLE=0 / 1 (forced off / on) must agree bit for bit on per-walk values,
    estimates, variances, in both stream forms and in the solve, duplicate diagonal entries included (the LAST stored match is the
    diagonal, as the scan finds it)"""
    n = 300
    rp, ci, va, _ = G.sdd_rows(n, 8, seed=5, half_bandwidth=40)
    rp, ci, va = np.array(rp), np.array(ci), np.array(va)
    k = int(rp[7])                                                      # row 7: a second stored entry at the diagonal's column
    ci[k] = 7 if ci[k] != 7 else ci[k]
    b = np.random.default_rng(8).standard_normal(n)
    m = S.SparseMatrix.from_csr(rp, ci, va, n, n, keep_csr=True)
    got = {}
    for tab in ("0", "1"):
        monkeypatch.setenv("SL_WALK_TABLE", tab)
        bv, br = _walk(m, b, 7, 700, 3)
        sv, sr = _walk(m, b, 7, 700, 3, stream=L.SL_WALK_STREAM_SERIAL)
        r1 = S.random_walk_solve(m, b, 0.1, 4)
        r2 = S.random_walk_solve(m, b, 0.1, 4, stream="reference")
        got[tab] = np.concatenate([bv, [br.estimate, br.variance], sv, [sr.estimate, sr.variance], r1["solution"], r1["variances"], r2["solution"],
                                   r2["variances"], [r1["total_variance"], r2["total_variance"]]]).view(np.uint64)
    assert (got["0"] == got["1"]).all()
    monkeypatch.delenv("SL_WALK_TABLE")
    ov, om, ovar = O.ts_random_walk_serial(rp, ci, va, b, 7, 700, 3)
    sv, sr = _walk(m, b, 7, 700, 3, stream=L.SL_WALK_STREAM_SERIAL)     # (the rule's own choice: 700 x 64 >= 300 rows -> table)
    assert (sv.view(np.uint64) == ov.view(np.uint64)).all() and (sr.estimate, sr.variance) == (om, ovar)


def test_block_stride_shrinks_beyond_the_generators_period(gpu):
    """more than 2^21 walks in one call: blocks of 1024 draws instead of a second pass over the same 2048-draw blocks — the
    per-walk values still equal the oracle's block form, and walk s no longer equals walk s + 2^21"""
    n = 64
    rp, ci, va, _ = G.sdd_rows(n, 4, seed=2)
    b = np.random.default_rng(6).standard_normal(n)
    m = S.SparseMatrix.from_csr(rp, ci, va, n, n, keep_csr=True)
    N = (1 << 21) + 4096
    gv, res = _walk(m, b, 5, N, 9)
    assert res.num_samples == N and O.walk_stride(N) == 1024
    head = np.arange(0, 4096)
    for s in (0, 1, 4095):                                                         # spot values against the block rule at THIS call's stride
        one, _, _ = O.ts_random_walk_streams(rp, ci, va, b, 5, 1, O.ts_lcg_jump(9, s * 1024))
        assert gv[s] == one[0]
    assert (gv[head] != gv[head + (1 << 21)]).any()
    with pytest.raises(S.SolverError) as e:
        _walk(m, b, 5, (1 << 28) + 1, 9)
    assert e.value.kind == "InvalidInput"
    with pytest.raises(S.SolverError) as e:                                         # an epsilon whose 1 / eps^2 is no integer any more: refused, not converted
        _walk(m, b, 5, 0, 9, eps=1e-12)
    assert e.value.kind == "InvalidInput" and "walks" in str(e.value)
