"""Every opt-in path of round 4 against the ORACLE itself — full arrays, not hashes, not "same as the default path" (VERDICT r04 item 2:
an opt-in becomes a default only after a direct comparison of x / r / frontier lists / push counts with oracle.push_sync_solve /
oracle.cg_solve; the self-comparisons of tests/test_gpu_session.py and tests/test_gpu_pagerank.py stay as the bit-for-bit A/B).

    SL_PUSH_SMALL=1       small push rounds back to back in one workgroup     forward_push.rs:67-216 (the data-parallel push)
    SL_QUERY_WIDE=W       W single-entry queries through one launch train      forward_push.rs:224-231
    SL_PW_INDEX_ONLY=1    index-only paced stream of column-constant operators (PageRank systems, core/solver.ts:664-722)
    SL_CG_FUSED_DOT=1     p.Ap inside the SpMV launch, merged vector pass      optimized_solver.rs:182-295

The switches are read once per process, so every case runs in a child process with the switch set; the child compares with the oracle
and prints one verdict line.  Bar: x, r, frontier index lists, rounds and push counts bit for bit; query estimates (sums over the touched
rows, tree order) to 1e-14 relative, as for the default path; CG to 1e-10 relative (its dots are tree-reduced), as tests/test_gpu_cg.py."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent

PRELUDE = r"""
import json, numpy as np
import sublinear_time_solver_amd as S
from sublinear_time_solver_amd import _lib as L
from sublinear_time_solver_amd import generators as G
from oracle import oracle as O
bad = []
def same_bits(a, b): return bool((np.ascontiguousarray(a, dtype=np.float64).view(np.uint64) == np.ascontiguousarray(b, dtype=np.float64).view(np.uint64)).all())
def check(what, ok):
    if not ok: bad.append(what)
def push_vs_oracle(tag, m, rp, ci, va, b, theta, dense_switch=1.0 / 16.0, cap=1 << 22, max_rounds=10_000):
    p = S.PushSolver(theta=theta, dense_switch=dense_switch, max_rounds=max_rounds).solve(m, b, log_frontier=cap)
    o = O.push_sync_solve(rp, ci, va, b, theta=theta, max_rounds=max_rounds, log_cap=cap)
    check(f"{tag}: rounds {p['rounds']} vs {o['rounds']}", p["rounds"] == o["rounds"])
    check(f"{tag}: pushes {p['pushes']} vs {o['pushes']}", p["pushes"] == o["pushes"])
    if p["dense_rounds"] == 0:      # (a dense round counts every row it walks; the oracle counts candidate rows)
        check(f"{tag}: rows touched", p["rows_touched"] == o["rows_touched"])
    check(f"{tag}: converged", p["converged"] == o["converged"])
    check(f"{tag}: frontier lists", p["frontier_log"].size == o["frontier_log"].size and bool((p["frontier_log"] == o["frontier_log"]).all()))
    check(f"{tag}: x bits", same_bits(p["solution"], o["x"]))
    check(f"{tag}: r bits", same_bits(p["residual"], o["r"]))
    return p
def query_vs_oracle(tag, e, trp, tci, tva, n, b, row, theta, max_rounds=100_000):
    seed = np.zeros(n); seed[row] = 1.0
    o = O.push_sync_solve(trp, tci, tva, seed, theta=theta, max_rounds=max_rounds)
    est, l1 = float(np.dot(o["x"], b)), float(np.abs(o["r"]).sum())
    check(f"{tag} row {row} theta {theta}: rounds {e.rounds} vs {o['rounds']}", e.rounds == o["rounds"])
    check(f"{tag} row {row} theta {theta}: pushes {e.pushes} vs {o['pushes']}", e.pushes == o["pushes"])
    check(f"{tag} row {row} theta {theta}: converged", bool(e.converged) == o["converged"])
    check(f"{tag} row {row} theta {theta}: estimate {e.estimate} vs {est}", abs(e.estimate - est) <= 1e-14 * max(1.0, abs(est)))
    check(f"{tag} row {row} theta {theta}: residual l1", abs(e.residual_l1 - l1) <= 1e-14 * max(1.0, l1))
"""


def _run(body, env, timeout=900):
    r = subprocess.run([sys.executable, "-c", PRELUDE + body + '\nprint("VERDICT " + json.dumps({"bad": bad, "info": info}))\n'], cwd=ROOT, capture_output=True,
                       text=True, timeout=timeout, env=dict(os.environ, **env))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    v = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("VERDICT ")][-1][8:])
    assert not v["bad"], v["bad"][:8]
    return v["info"], r.stderr


SMALL_ROUNDS = r"""
info = {}
n = 60_000
rp, ci, va, b = G.sdd_rows(n, 11, seed=9, half_bandwidth=0)
m = S.SparseMatrix.from_csr(rp, ci, va, n, n, with_transpose=True)
trp, tci, tva = O.csr_transpose(rp, ci, va, n)
rounds = []
with S.QuerySession(m, b) as q:
    for row, theta, mr in [(0, 1e-3, 100000), (n // 3, 1e-5, 100000), (n - 1, 1e-7, 100000), (5, 1e-6, 3), (0, 1e-3, 100000)]:
        e = q.estimate(row, theta=theta, max_rounds=mr)
        query_vs_oracle("sdd", e, trp, tci, tva, n, b, row, theta, mr)
        rounds.append(int(e.rounds))
info["query rounds"] = rounds
# sparse right-hand sides: frontiers that start small, grow through the launch train and die out small again
for k, (theta, stride) in enumerate([(1e-6, 997), (1e-8, 4001), (1e-4, 50)]):
    bs = b * (np.arange(n) % stride == 0)
    p = push_vs_oracle(f"sdd push {k}", m, rp, ci, va, bs, theta)
    info[f"sdd push {k}"] = [p["rounds"], p["pushes"], p["dense_rounds"]]
# a power-law graph: hub columns (long-column pieces), heavy rows
N = 20_000
adj = G.pagerank_graph(N, 3)
prp, pci, pva, pb = G.pagerank_system(N, *adj, damping=0.85)
pm = S.SparseMatrix.from_csr(prp, pci, pva, N, N, with_transpose=True)
ptr = O.csr_transpose(prp, pci, pva, N)
with S.QuerySession(pm, pb) as q:
    for row, theta in [(0, 1e-4), (7, 1e-6), (N - 1, 1e-5)]:
        query_vs_oracle("pagerank", q.estimate(row, theta=theta), *ptr, N, pb, row, theta)
p = push_vs_oracle("pagerank push", pm, prp, pci, pva, pb * (np.arange(N) % 97 == 0), 1e-7)
info["pagerank push"] = [p["rounds"], p["pushes"], p["dense_rounds"]]
"""


@pytest.mark.parametrize("small", ["1", "0"])
def test_small_rounds_kernel_against_the_oracle(gpu, small):
    info, _ = _run(SMALL_ROUNDS, {"SL_PUSH_SMALL": small})
    assert max(info["query rounds"]) >= 6 and info["sdd push 1"][0] >= 8          # several rounds did run, small frontiers among them


WIDE = r"""
info = {}
n = 50_000
rp, ci, va, b = G.sdd_rows(n, 9, seed=12, half_bandwidth=0)
m = S.SparseMatrix.from_csr(rp, ci, va, n, n, with_transpose=True)
tr = O.csr_transpose(rp, ci, va, n)
rows = [0, 17, n - 1, 4242, 17, 31337, 9, 25_000, 40_001, 3, 12_345]          # 11 = not a multiple of W, a row twice
with S.QuerySession(m, b) as q:
    for theta, mr in [(1e-4, 100000), (1e-6, 100000), (10.0, 100000), (1e-7, 3)]:
        got = q.estimate_batch(rows, theta=theta, max_rounds=mr)
        check("batch length", len(got) == len(rows))
        for row, e in zip(rows, got):
            query_vs_oracle(f"wide theta {theta} max_rounds {mr}", e, *tr, n, b, row, theta, mr)
    info["rounds at 1e-6"] = [int(e.rounds) for e in q.estimate_batch(rows, theta=1e-6)]
N = 20_000
adj = G.pagerank_graph(N, 5)
prp, pci, pva, pb = G.pagerank_system(N, *adj, damping=0.85)
pm = S.SparseMatrix.from_csr(prp, pci, pva, N, N, with_transpose=True)
ptr = O.csr_transpose(prp, pci, pva, N)
prow = [0, 5, N - 1, 777, 10_000, 64, 1]
with S.QuerySession(pm, pb) as q:
    for row, e in zip(prow, q.estimate_batch(prow, theta=1e-5)):
        query_vs_oracle("wide pagerank", e, *ptr, N, pb, row, 1e-5)
"""


@pytest.mark.parametrize("extra", [{}, {"SL_PUSH_SMALL": "1"}])
def test_wide_query_batches_against_the_oracle(gpu, extra):
    info, _ = _run(WIDE, dict(SL_QUERY_WIDE="4", **extra))
    assert max(info["rounds at 1e-6"]) >= 4


INDEX_ONLY = r"""
info = {}
n = 40_000
rp, ci, w = G.pagerank_graph(n, 11)
arp, aci, ava, b = G.pagerank_system(n, rp, ci, w, damping=0.85)
m = S.SparseMatrix.from_csr(arp, aci, ava, n, n, with_transpose=True)
info["layout"] = int(m.info().column_panels)
for theta in (1e-9, 1e-6):
    p = push_vs_oracle(f"unit graph theta {theta}", m, arp, aci, ava, b, theta, dense_switch=1.0 / 64.0, cap=1 << 24)
    info[f"dense rounds {theta}"] = int(p["dense_rounds"])
# a sparse start: the first rounds run the launch train, the middle ones the dense (index-only) kernel, the z vector must follow both
bs = b * (np.arange(n) % 211 == 0)
p = push_vs_oracle("unit graph, sparse start", m, arp, aci, ava, bs, 1e-10, dense_switch=1.0 / 64.0, cap=1 << 24)
info["dense rounds sparse start"] = int(p["dense_rounds"])
# weighted graph: not a column-constant operator — must run the full stream and still equal the oracle
w2 = 0.5 + (np.arange(w.size) % 7) * 0.25
wrp, wci, wva, wb = G.pagerank_system(n, rp, ci, w2, damping=0.85)
wm = S.SparseMatrix.from_csr(wrp, wci, wva, n, n, with_transpose=True)
push_vs_oracle("weighted graph", wm, wrp, wci, wva, wb, 1e-8, dense_switch=1.0 / 64.0, cap=1 << 24)
"""


@pytest.mark.parametrize("idx", ["1", "0"])
def test_index_only_stream_against_the_oracle(gpu, idx):
    info, log = _run(INDEX_ONLY, {"SL_PW_INDEX_ONLY": idx, "SL_COLUMN_PANELS": "1", "SL_PW_FORCE": "1", "SL_PW_CUS": "4", "SL_LOG": "1"}, timeout=1500)
    assert info["layout"] == 2 and info["dense rounds 1e-09"] > 3 and info["dense rounds sparse start"] > 0
    assert (log.count("column-constant operator") == 1) == (idx == "1")


CG = r"""
info = {}
import scipy.sparse as sp
def spd(n, k=7, seed=0):
    rng = np.random.default_rng(seed)
    B = sp.random(n, n, density=k / n, random_state=rng, data_rvs=lambda s: rng.uniform(-1, 1, s), format="csr")
    A = B + B.T
    d = np.abs(A).sum(axis=1).A1 + 1.0
    A = (A + sp.diags(d)).tocsr(); A.sort_indices()
    return A.indptr.astype(np.uint32), A.indices.astype(np.uint32), A.data.astype(np.float64)
for n, order in [(5000, 0), (5000, 1), (60000, 0), (200_003, 0)]:
    rp, ci, va = spd(n, seed=n)
    b = 1.0 + np.cos(np.arange(n))
    m = S.SparseMatrix.from_csr(rp, ci, va, n, n)
    g = S.ConjugateGradientSolver(tolerance=1e-7, order=order).solve(m, b)
    o = O.cg_solve(rp, ci, va, b, tolerance=1e-7, order=order)
    check(f"cg {n}/{order}: converged", bool(g.converged) and o["converged"])
    check(f"cg {n}/{order}: iterations {g.iterations} vs {o['iterations']}", abs(g.iterations - o["iterations"]) <= 1)
    if g.iterations == o["iterations"]:
        check(f"cg {n}/{order}: x to 1e-10", float(np.max(np.abs(g.solution - o["x"]))) <= 1e-10 * float(np.max(np.abs(o["x"]))))
    check(f"cg {n}/{order}: true residual", float(np.linalg.norm(O.spmv(rp, ci, va, g.solution) - b)) <= 2e-7)
    g2 = S.ConjugateGradientSolver(tolerance=1e-7, order=order).solve(m, b)
    check(f"cg {n}/{order}: run-to-run bits", same_bits(g.solution, g2.solution))
    info[f"{n}/{order}"] = [int(g.iterations), int(o["iterations"])]
# the reference's own known answer (optimized_solver.rs:401-420)
m = S.SparseMatrix.from_triplets([(0, 0, 4.0), (0, 1, 1.0), (1, 0, 1.0), (1, 1, 3.0)], 2, 2)
r = S.ConjugateGradientSolver().solve(m, [1.0, 2.0])
check("cg kat", bool(r.converged) and float(np.max(np.abs(r.solution - np.array([1.0 / 11.0, 7.0 / 11.0])))) <= 1e-12)
"""


@pytest.mark.parametrize("fused", ["1", "0"])
def test_cg_with_the_dot_in_the_row_kernel_against_the_oracle(gpu, fused):
    info, _ = _run(CG, {"SL_CG_FUSED_DOT": fused})
    assert all(abs(a - b) <= 1 for a, b in info.values())


ALL_ON = r"""
info = {}
n = 30_000
rp, ci, va, b = G.sdd_rows(n, 16, seed=21, half_bandwidth=0)
m = S.SparseMatrix.from_csr(rp, ci, va, n, n, with_transpose=True)
g = S.NeumannSolver().solve(m, b, S.SolverOptions(tolerance=1e-10))
o = O.neumann_solve(rp, ci, va, b, tolerance=1e-10)
check("neumann iterations", g.iterations == o["iterations"])
check("neumann x bits", same_bits(g.solution, o["x"]))
push_vs_oracle("push, every switch on", m, rp, ci, va, b * (np.arange(n) % 501 == 0), 1e-9)
"""


def test_every_switch_on_at_once_leaves_the_default_solves_alone(gpu):
    _run(ALL_ON, {"SL_PUSH_SMALL": "1", "SL_QUERY_WIDE": "4", "SL_CG_FUSED_DOT": "1", "SL_PW_INDEX_ONLY": "1"})
