#!/usr/bin/env python3
"""BASELINE configs[1] (n = 10^6, 8 per row, full solve to 1e-8) a few times, for `rocprofv3 --kernel-trace --stats`: which launches
the 1.1 ms of the solve loop are made of."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sublinear_time_solver_amd as S
from sublinear_time_solver_amd import generators as G
w = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = 1_000_000
rp, ci, va, b = G.sdd_rows(n, 8, seed=1, half_bandwidth=w)
m = S.SparseMatrix.from_csr(rp, ci, va, n, n)
print("layout", m.info().column_panels)
sol = S.NeumannSolver()
for _ in range(5):
    r = sol.solve(m, b, S.SolverOptions(tolerance=1e-8, collect_stats=True))
print(r.iterations, r.stats["matvec_count"], r.stats["device_time_ms"])
