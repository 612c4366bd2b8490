#!/usr/bin/env python3
"""G9 — golden vectors for the PageRank system of the shipped surface (`computePageRank`, src/core/solver.ts:664-722: solve
(I - d P^T) x = (1 - d)/n with P the row-normalised adjacency, rows without out-links left empty) from the reference's own
runnable Python power iteration `SublinearPageRank._simple_pagerank` (scripts/pagerank/sublinear_pagerank.py:145-166), which
iterates exactly that fixed point.  Seeded digraphs with dangling nodes, self loops and weights; epsilon 1e-15 so that the
iteration runs to its floating-point fixed point.

Runs ONLY in the build container (imports the reference module from /root/reference with bytecode writing off).  Only derived
data — the adjacency matrices (as COO) and the returned vectors — is written to tests/golden/reference_pagerank.npz."""
import os
import sys
from pathlib import Path

import numpy as np

OUT = Path(__file__).resolve().parent
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/scripts/pagerank")
from sublinear_pagerank import SublinearPageRank  # noqa: E402


def digraph(n, density, seed, weighted, dangling):
    rng = np.random.default_rng(seed)
    A = (rng.random((n, n)) < density).astype(np.float64)
    if weighted:
        A *= rng.integers(1, 5, size=(n, n))
    for i in dangling:
        A[i, :] = 0.0                     # no out-links
    return A


def main():
    cases = {"ring5": np.roll(np.eye(5), 1, axis=1) + np.diag([0, 0, 1.0, 0, 0]),
             "dense30": digraph(30, 0.3, 1, False, []),
             "sparse200": digraph(200, 0.03, 2, False, [3, 77, 150]),
             "weighted120": digraph(120, 0.06, 3, True, [0, 119]),
             "hub150": digraph(150, 0.02, 4, False, [10])}
    cases["hub150"][:, 5] = 1.0           # everybody links to node 5
    cases["hub150"][10, :] = 0.0
    store, names = {}, []
    for name, A in cases.items():
        for d in (0.85, 0.5):
            pr = SublinearPageRank(damping=d, epsilon=1e-15, max_iterations=5000)._simple_pagerank(A.copy())
            r, c = np.nonzero(A)
            key = f"{name}__d{int(d * 100)}"
            store[f"{key}__rows"], store[f"{key}__cols"], store[f"{key}__vals"] = r.astype(np.uint32), c.astype(np.uint32), A[r, c]
            store[f"{key}__n"] = np.asarray([A.shape[0]])
            store[f"{key}__damping"] = np.asarray([d])
            store[f"{key}__pagerank"] = np.asarray(pr, dtype=np.float64)
            names.append(key)
    store["__cases"] = np.asarray(names)
    np.savez_compressed(OUT / "reference_pagerank.npz", **store)
    print("wrote", OUT / "reference_pagerank.npz", os.path.getsize(OUT / "reference_pagerank.npz"), "bytes;", len(names), "cases")


if __name__ == "__main__":
    main()
