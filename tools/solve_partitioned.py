#!/usr/bin/env python3
"""BASELINE config 5 as a SOLVE: S-DD(n_global = ranks x --rows, k, w) row-partitioned over the ranks of one node, every
rank building its own row slice in HBM, solved to --tolerance by distributed.PartitionedNeumannSolver (the control flow of
NeumannSolver::solve, neumann.rs:469-555: halo / all-gather exchange of the term each iteration and of the solution at each
residual check, all-reduced norms).  One process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \\
        tools/solve_partitioned.py --rows 10000000

`--exchange abi` runs the SAME solve through the library's own communicator and partitioned NeumannState (C ABI: sl_comm_*,
sl_neumann_state_create_partitioned / _run / _solution; no process group is initialised, ranks may share a GPU).
SL_BENCH_BACKEND=gloo stages the torch.distributed exchanges through the host so that several ranks can share one GPU (tests).
Rank 0 prints one JSON object."""
import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # RCCL peer access on this platform needs dmabuf IPC


def main_abi(a, world, rank, local_rank, dev):
    """the same solve through distributed.AbiPartitionedNeumannSolver: communicator + partitioned NeumannState of the C ABI"""
    import torch
    from sublinear_time_solver_amd import _lib as L
    from sublinear_time_solver_amd import distributed as D
    lib = L.load()
    L.check(lib.sl_set_device(local_rank))
    k, w = a.k, a.bandwidth
    if a.bounds:
        bounds = [int(v) for v in a.bounds.split(",")]
        n_global = bounds[-1]
        part = D.RowPartition(n_global, world, rank, bounds=bounds)
    else:
        n_global = a.rows * world
        part = D.RowPartition(n_global, world, rank)
    n_local = part.n_local
    rp = torch.empty(n_local + 1, dtype=torch.int32, device=dev)
    ci = torch.empty(n_local * k, dtype=torch.int32, device=dev)
    va = torch.empty(n_local * k, dtype=torch.float64, device=dev)
    b = torch.empty(n_local, dtype=torch.float64, device=dev)
    L.check(lib.sl_synth_sdd_device(n_global, k, a.seed, w, part.lo, part.hi, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), b.data_ptr()))
    h = C.c_void_p()
    L.check(lib.sl_matrix_create_csr(n_local, n_global, n_local * k, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), L.SL_MEM_DEVICE, part.lo, 0, C.byref(h)))
    del rp, ci, va
    solver = D.AbiPartitionedNeumannSolver(rank, world, f"solve_{os.environ.get('MASTER_PORT', '0')}")
    solver.comm.barrier()
    t0 = time.perf_counter()
    r = solver.solve(h, b, tolerance=a.tolerance, order=a.order)
    solver.comm.barrier()
    dt = time.perf_counter() - t0
    x = r["solution_local"]
    sx = sum(solver.comm.allgather_f64(float(x.sum())))
    sx2 = sum(solver.comm.allgather_f64(float((x * x).sum())))
    if rank == 0:
        launches = r["terms"] - 1 + (r["iterations"] + 4) // 5 + 1
        print(json.dumps({"config": f"S-DD(n={n_global}, nnz/row={k}, w={w}) over {world} rank(s), exchange abi (sl_comm)", "n_gpus": world,
                          "iterations": r["iterations"], "terms": r["terms"], "converged": r["converged"], "residual_norm": r["residual_norm"],
                          "solve_s": dt, "rows_iter_per_s": n_global * launches / dt, "nnz_iter_per_s": n_global * k * launches / dt,
                          "sum_x": sx, "sum_x2": sx2, "last_term_norm": r["last_term_norm"]}), flush=True)
    solver.close()
    lib.sl_matrix_destroy(h)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000, help="rows per rank")
    ap.add_argument("--k", type=int, default=16)
    ap.add_argument("--bandwidth", type=int, default=4096)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--tolerance", type=float, default=1e-8)
    ap.add_argument("--order", type=int, default=0)
    ap.add_argument("--exchange", choices=["torch", "abi"], default="torch", help="torch = torch.distributed (RCCL / gloo); abi = sl_comm behind the C ABI")
    ap.add_argument("--bounds", type=str, default="", help="explicit row bounds b0,b1,...,bN (N = ranks, b0 = 0, bN = n_global) instead of "
                    "--rows per rank: unequal row ranges, as nnz-balanced bounds of a ragged system would be")
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    from sublinear_time_solver_amd import _lib as L
    from sublinear_time_solver_amd import distributed as D

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("SL_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(1, torch.cuda.device_count())
    if a.exchange == "abi":
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if a.exchange == "abi":
        return main_abi(a, world, rank, local_rank, dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    lib = L.load()
    L.check(lib.sl_set_device(local_rank))
    L.check(lib.sl_set_stream(C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))

    k, w = a.k, a.bandwidth
    if a.bounds:
        bounds = [int(v) for v in a.bounds.split(",")]
        n_global = bounds[-1]
        part = D.RowPartition(n_global, world, rank, bounds=bounds)
    else:
        n_global = a.rows * world
        part = D.RowPartition(n_global, world, rank)
    n_local = part.n_local
    rp = torch.empty(n_local + 1, dtype=torch.int32, device=dev)
    ci = torch.empty(n_local * k, dtype=torch.int32, device=dev)
    va = torch.empty(n_local * k, dtype=torch.float64, device=dev)
    b = torch.empty(n_local, dtype=torch.float64, device=dev)
    L.check(lib.sl_synth_sdd_device(n_global, k, a.seed, w, part.lo, part.hi, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), b.data_ptr()))
    h = C.c_void_p()
    L.check(lib.sl_matrix_create_csr(n_local, n_global, n_local * k, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), L.SL_MEM_DEVICE, part.lo, 0, C.byref(h)))
    del rp, ci, va
    dinv = torch.empty(n_local, dtype=torch.float64, device=dev)
    L.check(lib.sl_matrix_diagonal_inverse(h, dinv.data_ptr(), L.SL_MEM_DEVICE))
    ops = D.hip_local_ops(h, dinv, a.order)
    ex = D.HaloExchange(part, w) if w else D.AllGatherExchange(part)
    solver = D.PartitionedNeumannSolver(part, ops, ex)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    r = solver.solve(b, dinv, tolerance=a.tolerance)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    checksum = torch.zeros(2, dtype=torch.float64, device=dev)
    checksum[0] = r.solution_local.sum()
    checksum[1] = (r.solution_local * r.solution_local).sum()
    if world > 1:
        D.all_reduce_scalar(checksum, dist.ReduceOp.SUM)
    if rank == 0:
        launches = r.terms_computed - 1 + (r.iterations + 4) // 5 + 1          # fused steps + residual checks
        print(json.dumps({"config": f"S-DD(n={n_global}, nnz/row={k}, w={w}) over {world} rank(s), exchange {ex.name}", "n_gpus": world,
                          "iterations": r.iterations, "terms": r.terms_computed, "converged": bool(r.converged), "residual_norm": r.residual_norm,
                          "solve_s": dt, "rows_iter_per_s": n_global * launches / dt, "nnz_iter_per_s": n_global * k * launches / dt,
                          "sum_x": float(checksum[0]), "sum_x2": float(checksum[1]), "last_term_norm": r.term_norms[-1]}), flush=True)
    lib.sl_matrix_destroy(h)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
