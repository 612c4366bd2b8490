#!/usr/bin/env python3
"""BASELINE configs 0 and 1 end to end on one MI355X: full solves through the drop-in boundary, timed three ways —
 (1) device time of the solve loop (inputs resident in HBM: `device_time_ms` of sl_neumann_solve),
 (2) wall time of the sl_neumann_solve call with HOST b / x (adds the vector transfers and host readbacks),
 (3) wall time including sl_matrix_create_csr from HOST CSR arrays (the PCIe-inclusive figure DESIGN.md quotes),
next to the CPU restatement of the reference on this box's host cores (one thread, the reference's own loop).
Writes one JSON object to stdout.  Uses oracle/ only for the CPU timing leg and the residual cross-check."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import sublinear_time_solver_amd as S                     # noqa: E402
from sublinear_time_solver_amd import generators as G     # noqa: E402


def run_case(name, rp, ci, va, b, tol, cpu=True, repeats=3):
    n = b.size
    out = {"case": name, "n": int(n), "nnz": int(va.size), "tolerance": tol}
    creates = []
    for _ in range(3):                                    # steady state: the first create of a PROCESS also pays the runtime's start-up (main())
        t0 = time.perf_counter()
        m = S.SparseMatrix.from_csr(rp, ci, va, n, n)
        creates.append((time.perf_counter() - t0) * 1e3)
    out["create_from_host_ms"] = min(creates)
    out["create_from_host_ms_all"] = creates
    sol = S.NeumannSolver()
    best = None
    for _ in range(repeats):
        t0 = time.perf_counter()
        r = sol.solve(m, b, S.SolverOptions(tolerance=tol, collect_stats=True))
        wall = (time.perf_counter() - t0) * 1e3
        if best is None or wall < best[0]:
            best = (wall, r)
    wall, r = best
    nnz_iter = float(r.stats["matvec_count"]) * va.size
    out.update(iterations=r.iterations, converged=bool(r.converged), residual_norm=r.residual_norm,
               matvecs=int(r.stats["matvec_count"]), solve_device_ms=r.stats["device_time_ms"], solve_call_wall_ms=wall,
               create_plus_solve_ms=out["create_from_host_ms"] + wall,
               nnz_iter_per_s_device=nnz_iter / (r.stats["device_time_ms"] * 1e-3),
               nnz_iter_per_s_call=nnz_iter / (wall * 1e-3),
               nnz_iter_per_s_pcie_inclusive=nnz_iter / ((out["create_from_host_ms"] + wall) * 1e-3),
               host_bytes_in=int(va.nbytes + ci.nbytes + rp.nbytes + b.nbytes), host_bytes_out=int(b.nbytes))
    true_res = np.linalg.norm(b - _spmv(rp, ci, va, r.solution))
    out["residual_recomputed_on_host"] = float(true_res)
    if cpu:
        from oracle import oracle as O
        t0 = time.perf_counter()
        o = O.neumann_solve(rp, ci, va, b, tolerance=tol, fast=True)
        out["cpu_port_solve_ms"] = (time.perf_counter() - t0) * 1e3
        out["cpu_iterations"] = int(o["iterations"])
        out["solution_bits_equal_cpu"] = bool((o["x"].view(np.uint64) == r.solution.view(np.uint64)).all())
        out["speedup_device_vs_cpu_1thread"] = out["cpu_port_solve_ms"] / out["solve_device_ms"]
    return out


def _spmv(rp, ci, va, x):
    import scipy.sparse as sp
    return sp.csr_matrix((va, ci.astype(np.int64), rp.astype(np.int64)), shape=(rp.size - 1, x.size)) @ x


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--k", type=int, default=8)
    ap.add_argument("--bandwidth", type=int, default=0, help="0 = columns uniform over the matrix (the reference recipe)")
    ap.add_argument("--tolerance", type=float, default=1e-8)
    args = ap.parse_args()
    res = []
    # what the first call of a process pays once, whatever its size: HIP runtime and code-object load, the first allocations, the
    # sort library's first launch — measured on a 64-row matrix and reported apart from the configurations
    rp0, ci0, va0, b0 = G.sdd_rows(64, 4, seed=1)
    t0 = time.perf_counter()
    m0 = S.SparseMatrix.from_csr(rp0, ci0, va0, 64, 64)
    first_create = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    S.NeumannSolver().solve(m0, b0, S.SolverOptions(tolerance=1e-8))
    first_solve = (time.perf_counter() - t0) * 1e3
    init = {"first_matrix_create_of_the_process_ms": first_create, "first_solve_of_the_process_ms": first_solve}
    # config 0: the TS `generate -t diagonally-dominant -s 1000` recipe (dense 1000 x 1000)
    rp, ci, va, b = G.gen1000_dense(1000, seed=12345)
    res.append(run_case("config0: n=1000 generate -t diagonally-dominant", rp, ci, va, b, 1e-10))
    # config 1: n = 1M, 8 nnz/row, full solve to 1e-8
    for w in sorted({args.bandwidth, 1024}):
        rp, ci, va, b = G.sdd_rows(args.n, args.k, seed=1, half_bandwidth=w)
        res.append(run_case(f"config1: n={args.n} nnz/row={args.k} S-DD w={w} full solve", rp, ci, va, b, args.tolerance))
    print(json.dumps({"process_init": init, "full_solves": res}, indent=1))


if __name__ == "__main__":
    main()
