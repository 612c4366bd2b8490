#!/usr/bin/env python3
"""BASELINE config 4: n = 10M PageRank (alpha = 0.85) power-law graph, single-entry estimateEntry() query, 1xMI355X.

Builds S-PR(n, seed) in HBM as M = I - alpha P (row-stochastic side), i.e. the transpose of the PageRank solve
matrix A = I - alpha P^T (src/core/solver.ts:664-722).  A query x_row = (A^-1 b)_row is a local push on A^T = M
from e_row (sl_estimate_entry_transposed).  Checked against the full solve of A x = b (thresholded push on
sl_matrix_transpose(M)) through the error bound |x_row - estimate| <= ||r_y||_1 * ||x||_inf.
"""
import argparse
import ctypes as C
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--alpha", type=float, default=0.85)
    ap.add_argument("--thetas", type=str, default="1e-5,1e-7,1e-9")
    ap.add_argument("--no-full-solve", action="store_true")
    args = ap.parse_args()
    import os
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")          # the lanes of the batch entry overlap on the device up to the runtime's hardware queues
    import torch
    import sublinear_time_solver_amd as S
    from sublinear_time_solver_amd import _lib as L
    lib = L.load()
    n = args.n
    dev = torch.device("cuda", 0)
    t0 = time.perf_counter()
    rp = torch.empty(n + 1, dtype=torch.int32, device=dev)
    nnz = C.c_uint64(0)
    L.check(lib.sl_synth_pagerank_device(n, args.seed, args.alpha, 2, 8192, rp.data_ptr(), None, None, C.byref(nnz)))
    ci = torch.empty(nnz.value, dtype=torch.int32, device=dev)
    va = torch.empty(nnz.value, dtype=torch.float64, device=dev)
    L.check(lib.sl_synth_pagerank_device(n, args.seed, args.alpha, 2, 8192, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), C.byref(nnz)))
    M = S.SparseMatrix.from_csr(rp, ci, va, n, n, with_transpose=True, device=True)
    L.check(lib.sl_synchronize())
    t_build = time.perf_counter() - t0
    b = torch.full((n,), (1.0 - args.alpha) / n, dtype=torch.float64, device=dev)
    info = M.info()
    # query rows: highest in-degree node (low ids attract links), a median node, a leaf-ish node
    rows = [0, n // 2, n - 1]
    out = {"config": f"S-PR(n={n}, seed={args.seed}, alpha={args.alpha})", "nnz": int(info.nnz), "build_s": t_build,
           "max_out_degree": int(info.max_row_nnz) - 1, "queries": []}
    x = None
    if not args.no_full_solve:
        t1 = time.perf_counter()
        A = M.transpose(with_transpose=True)
        o = L.PushOptions(); lib.sl_push_options_default(C.byref(o))
        o.theta, o.max_rounds, o.mem = 1e-13 / n, 10000, L.SL_MEM_DEVICE
        x = torch.zeros(n, dtype=torch.float64, device=dev)
        r = torch.empty(n, dtype=torch.float64, device=dev)
        res = L.PushResult()
        L.check(lib.sl_push_solve(A._h, b.data_ptr(), C.byref(o), x.data_ptr(), r.data_ptr(), None, 0, None, C.byref(res)))
        out["full_solve"] = {"rounds": int(res.rounds), "dense_rounds": int(res.dense_rounds), "residual_l2": res.residual_norm,
                             "device_ms": res.device_time_ms, "wall_s": time.perf_counter() - t1, "sum_x": float(x.sum()),
                             "padded_over_nnz": A.info().padded_nnz / A.info().nnz}
    xinf = float(x.abs().max()) if x is not None else None
    # one-shot queries pay the O(n) setup each time; a session pays it once
    t2 = time.perf_counter()
    e = S.estimate_entry(M, b, rows[0], theta=1e-5, max_rounds=100000, matrix_is_transpose=True, device=True)
    out["one_shot_query_wall_ms"] = (time.perf_counter() - t2) * 1e3
    t2 = time.perf_counter()
    sess = S.QuerySession(M, b, matrix_is_transpose=True, device=True)
    out["session_create_ms"] = (time.perf_counter() - t2) * 1e3
    sess.estimate(rows[1], theta=1e-5)                       # warm-up (first use of the kernels)
    for row, theta in [(r, float(t)) for r in rows for t in args.thetas.split(",")]:
        t2 = time.perf_counter()
        e = sess.estimate(row, theta=theta, max_rounds=100000)
        q = {"row": row, "theta": theta, "estimate": e.estimate, "residual_l1": e.residual_l1, "rounds": int(e.rounds), "pushes": int(e.pushes),
             "rows_touched": int(e.rows_touched), "touched_per_round_over_n": e.rows_touched / max(1, e.rounds) / n,
             "device_ms": e.device_time_ms, "wall_ms": (time.perf_counter() - t2) * 1e3, "converged": bool(e.converged)}
        if x is not None:
            q["full_solve_value"] = float(x[row])
            q["abs_error"] = abs(e.estimate - float(x[row]))
            q["error_bound"] = e.residual_l1 * xinf
            q["within_bound"] = bool(q["abs_error"] <= q["error_bound"] + 1e-18)
        out["queries"].append(q)
    # throughput of a stream of distinct local queries on one session
    import random
    rnd = random.Random(7)
    qrows = [rnd.randrange(n) for _ in range(200)]
    t2 = time.perf_counter()
    touched = 0
    for row in qrows:
        touched += sess.estimate(row, theta=1e-5).rows_touched
    dt = time.perf_counter() - t2
    out["query_stream"] = {"queries": len(qrows), "theta": 1e-5, "mean_wall_ms": dt / len(qrows) * 1e3, "queries_per_s": len(qrows) / dt,
                           "mean_rows_touched": touched / len(qrows)}
    # the same kind of stream through the batch entry: K queries at once on lanes (own state + stream + host thread each)
    out["query_batches"] = []
    for lanes in (1, 2, 4, 8, 16):
        for K in (16, 256):
            rows_k = [rnd.randrange(n) for _ in range(K)]
            sess.estimate_batch(rows_k[: min(K, lanes)], theta=1e-5, lanes=lanes)          # lanes set up / warm
            t2 = time.perf_counter()
            res = sess.estimate_batch(rows_k, theta=1e-5, lanes=lanes)
            dt = time.perf_counter() - t2
            out["query_batches"].append({"K": K, "lanes": lanes, "theta": 1e-5, "queries_per_s": K / dt, "mean_wall_ms_per_query": dt / K * 1e3,
                                         "mean_rows_touched": sum(r.rows_touched for r in res) / K})
    sess.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
