#!/usr/bin/env python3
"""Kernel-level view of local queries (run under rocprofv3 --kernel-trace --stats): S-PR graph, one session,
`--reps` repetitions of one (row, theta) query.  Prints wall time per query."""
import argparse
import ctypes as C
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--row", type=int, default=-1)
    ap.add_argument("--theta", type=float, default=1e-5)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    import torch
    import sublinear_time_solver_amd as S
    from sublinear_time_solver_amd import _lib as L
    lib = L.load()
    n = args.n
    row = args.row if args.row >= 0 else n - 1
    dev = torch.device("cuda", 0)
    rp = torch.empty(n + 1, dtype=torch.int32, device=dev)
    nnz = C.c_uint64(0)
    L.check(lib.sl_synth_pagerank_device(n, 1, 0.85, 2, 8192, rp.data_ptr(), None, None, C.byref(nnz)))
    ci = torch.empty(nnz.value, dtype=torch.int32, device=dev)
    va = torch.empty(nnz.value, dtype=torch.float64, device=dev)
    L.check(lib.sl_synth_pagerank_device(n, 1, 0.85, 2, 8192, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), C.byref(nnz)))
    M = S.SparseMatrix.from_csr(rp, ci, va, n, n, with_transpose=True, device=True)
    b = torch.full((n,), 0.15 / n, dtype=torch.float64, device=dev)
    with S.QuerySession(M, b, matrix_is_transpose=True, device=True) as q:
        q.estimate(row, theta=args.theta)
        t0 = time.perf_counter()
        for _ in range(args.reps):
            e = q.estimate(row, theta=args.theta)
        dt = (time.perf_counter() - t0) / args.reps
    print(f"row {row} theta {args.theta}: rounds {e.rounds} pushes {e.pushes} rows_touched {e.rows_touched} device_ms {e.device_time_ms:.3f} wall_ms {dt * 1e3:.3f}")


if __name__ == "__main__":
    main()
