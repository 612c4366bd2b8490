"""CLI-shaped front end over the C ABI: the `solve | analyze | pagerank | generate`
commands and flags of the reference's TS CLI (src/cli/index.ts:56-352), printing the same result fields.

    python -m sublinear_time_solver_amd solve -m A.json -b b.json --method neumann --epsilon 1e-6 [-o x.json]
    python -m sublinear_time_solver_amd analyze -m A.mtx
    python -m sublinear_time_solver_amd pagerank -g adjacency.json --damping 0.85 [--top 10]
    python -m sublinear_time_solver_amd generate -t diagonally-dominant -s 1000 -o A.json
"""
from __future__ import annotations

import argparse
import json
import sys
import time

import numpy as np


def _solve(a):
    from . import io, solver
    matrix = io.load_matrix(a.matrix)
    vector = io.load_vector(a.vector)
    print(f"Matrix: {matrix['rows']}x{matrix['cols']} ({matrix['format']})")
    print(f"Vector: length {len(vector)}")
    analysis = io.analyze_matrix(matrix)
    if a.verbose:
        print("Matrix Analysis:")
        print(f"  Diagonally dominant: {analysis['isDiagonallyDominant']}")
        print(f"  Dominance type: {analysis['dominanceType']}")
        print(f"  Dominance strength: {analysis['dominanceStrength']:.4f}")
        print(f"  Symmetric: {analysis['isSymmetric']}")
        print(f"  Sparsity: {analysis['sparsity'] * 100:.1f}%")
    if not analysis["isDiagonallyDominant"]:
        print("Warning: Matrix is not diagonally dominant. Convergence not guaranteed.", file=sys.stderr)
    print(f"Solving with method: {a.method}")
    print(f"Tolerance: {a.epsilon}")
    t0 = time.perf_counter()
    method = a.method
    if method == "neumann" and analysis["dominanceType"] == "column":
        method = "forward-push"        # the Rust Neumann path needs ROW dominance (matrix/mod.rs:467-485); push does not
    push = method != "neumann"
    m = io.matrix_to_device(matrix, with_transpose=push)
    res = solver.SublinearSolver(method=method, epsilon=a.epsilon, max_iterations=a.max_iterations).solve(m, vector)
    elapsed = (time.perf_counter() - t0) * 1e3
    print("\nSolution completed!")
    print(f"  Converged: {res['converged']}")
    print(f"  Iterations: {res['iterations']}")
    print(f"  Residual: {res['residual']:.6e}")
    print(f"  Time: {elapsed:.1f}ms")
    if a.output:
        with open(a.output, "w") as f:
            json.dump({"solution": np.asarray(res["solution"]).tolist(), "iterations": res["iterations"], "residual": res["residual"],
                       "converged": res["converged"], "method": res["method"], "computeTime": res["computeTime"]}, f)
        print(f"Solution written to {a.output}")
    else:
        sol = np.asarray(res["solution"])
        print("  Solution (first 10):", np.array2string(sol[:10], precision=6))
    return 0


def _analyze(a):
    from . import io
    matrix = io.load_matrix(a.matrix)
    print(json.dumps(io.analyze_matrix(matrix), indent=2))
    return 0


def _pagerank(a):
    """computePageRank (src/core/solver.ts:664-722): S = I - d P^T, rhs (1-d)/n (or personalised)."""
    from . import io, generators as G, solver
    adj = io.load_matrix(a.graph)
    r, c, v, rows, cols = io.matrix_to_triplets(adj)
    if rows != cols:
        raise SystemExit("adjacency matrix must be square")
    arp, aci, ava = G.adjacency_csr_first_match(r, c, v, rows)
    rp, ci, va, b = G.pagerank_system(rows, arp, aci, ava, a.damping)
    m = solver.SparseMatrix.from_csr(rp, ci, va, rows, rows, with_transpose=True)
    res = solver.PushSolver(theta=a.epsilon / rows, max_rounds=a.max_iterations).solve(m, b)
    x = res["solution"]
    order = np.argsort(-x)[: a.top]
    print(json.dumps({"converged": res["converged"], "iterations": res["rounds"], "residual": res["residual_norm"],
                      "totalScore": float(x.sum()), "topNodes": [{"node": int(i), "score": float(x[i])} for i in order]}, indent=2))
    return 0


def _generate(a):
    from . import io
    m = io.generate_matrix(a.type, a.size, seed=a.seed)
    out = json.dumps(m)
    if a.output:
        with open(a.output, "w") as f:
            f.write(out)
        print(f"Matrix written to {a.output}")
    else:
        print(out)
    return 0


def main(argv=None):
    p = argparse.ArgumentParser(prog="sublinear_time_solver_amd", description="MI355X-native sublinear solver front end")
    sub = p.add_subparsers(dest="cmd", required=True)
    s = sub.add_parser("solve", help="Solve a linear system from files")
    s.add_argument("-m", "--matrix", required=True)
    s.add_argument("-b", "--vector", required=True)
    s.add_argument("-o", "--output")
    s.add_argument("--method", default="neumann")
    s.add_argument("--epsilon", type=float, default=1e-6)
    s.add_argument("--max-iterations", type=int, default=1000)
    s.add_argument("--timeout", type=int)
    s.add_argument("--verbose", action="store_true")
    s.set_defaults(fn=_solve)
    an = sub.add_parser("analyze", help="Analyze matrix properties")
    an.add_argument("-m", "--matrix", required=True)
    an.set_defaults(fn=_analyze)
    pr = sub.add_parser("pagerank", help="Compute PageRank of a graph given as adjacency matrix")
    pr.add_argument("-g", "--graph", required=True)
    pr.add_argument("--damping", type=float, default=0.85)
    pr.add_argument("--epsilon", type=float, default=1e-6)
    pr.add_argument("--max-iterations", type=int, default=1000)
    pr.add_argument("--top", type=int, default=10)
    pr.set_defaults(fn=_pagerank)
    g = sub.add_parser("generate", help="Generate a test matrix")
    g.add_argument("-t", "--type", default="diagonally-dominant")
    g.add_argument("-s", "--size", type=int, default=100)
    g.add_argument("-o", "--output")
    g.add_argument("--seed", type=int, default=42)
    g.set_defaults(fn=_generate)
    a = p.parse_args(argv)
    from ._lib import SolverError
    try:
        return a.fn(a)
    except SolverError as e:
        print(f"Error: {e}", file=sys.stderr)
        return 1


if __name__ == "__main__":
    sys.exit(main())
