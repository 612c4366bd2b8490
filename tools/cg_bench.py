#!/usr/bin/env python3
"""f-1 (CG behind the same SpMV) at size: a 7-point stencil on an m x m x m grid (symmetric, strictly dominant: diagonal 6 + shift), solved
by sl_cg_solve with device-resident b / x; reports iterations, device time per iteration and the fraction of the 8 TB/s roofline on the
algorithmic bytes of one iteration — A p (12 nnz + 4 n + 16 n), p.Ap (16 n), x / r update with r.r (48 n), new direction (24 n).
One JSON line.  usage: python tools/cg_bench.py [--m 215] [--shift 0.05] [--tol 1e-8]"""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sublinear_time_solver_amd import _lib as L


def stencil(m, shift):
    n = m ** 3
    idx = np.arange(n, dtype=np.int64)
    i, j, k = idx // (m * m), (idx // m) % m, idx % m
    cols = [idx]
    vals = [np.full(n, 6.0 + shift)]
    keep = [np.ones(n, dtype=bool)]
    for d, c in ((-m * m, i > 0), (-m, j > 0), (-1, k > 0), (1, k < m - 1), (m, j < m - 1), (m * m, i < m - 1)):
        cols.append(idx + d); vals.append(np.full(n, -1.0)); keep.append(c)
    cols, vals, keep = np.stack(cols, 1), np.stack(vals, 1), np.stack(keep, 1)
    order = np.argsort(np.where(keep, cols, 1 << 62), axis=1, kind="stable")
    cols, vals, keep = np.take_along_axis(cols, order, 1), np.take_along_axis(vals, order, 1), np.take_along_axis(keep, order, 1)
    lens = keep.sum(1)
    rp = np.zeros(n + 1, dtype=np.uint32)
    rp[1:] = np.cumsum(lens)
    return rp, cols[keep].astype(np.uint32), vals[keep], n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=215)
    ap.add_argument("--shift", type=float, default=0.05)
    ap.add_argument("--tol", type=float, default=1e-8)
    args = ap.parse_args()
    lib = L.load()
    dev = torch.device("cuda", 0)
    rp, ci, va, n = stencil(args.m, args.shift)
    nnz = int(va.size)
    h = C.c_void_p()
    L.check(lib.sl_matrix_create_csr(n, n, nnz, L.ptr(rp), L.ptr(ci), L.ptr(va), L.SL_MEM_HOST, 0, 0, C.byref(h)))
    b = torch.ones(n, dtype=torch.float64, device=dev) + 0.001 * torch.remainder(torch.arange(n, device=dev, dtype=torch.float64), 1000.0)
    x = torch.empty(n, dtype=torch.float64, device=dev)
    o = L.CgOptions()
    lib.sl_cg_options_default(C.byref(o))
    o.tolerance, o.max_iterations, o.mem = args.tol, 5000, L.SL_MEM_DEVICE
    best = None
    for _ in range(3):
        r = L.CgResult()
        L.check(lib.sl_cg_solve(h, C.c_void_p(b.data_ptr()), C.byref(o), C.c_void_p(x.data_ptr()), C.byref(r)))
        if best is None or r.device_time_ms < best[0]:
            best = (r.device_time_ms, int(r.iterations), int(r.matvec_count), r.residual_norm, bool(r.converged))
    ms, it, mv, resn, conv = best
    xh = x.cpu().numpy()
    import scipy.sparse as sp
    a = sp.csr_matrix((va, ci.astype(np.int64), rp.astype(np.int64)), shape=(n, n))
    true_res = float(np.linalg.norm(b.cpu().numpy() - a @ xh))
    per = ms / max(mv, 1)
    bytes_it = 12 * nnz + 4 * (n + 1) + 16 * n + 16 * n + 48 * n + 24 * n
    lib.sl_matrix_destroy(h)
    print(json.dumps({"case": f"7-point stencil {args.m}^3, diagonal 6 + {args.shift}", "n": n, "nnz": nnz, "tolerance": args.tol, "iterations": it, "matvecs": mv,
                      "converged": conv, "residual_norm_reported": resn, "residual_recomputed_on_host": true_res, "device_ms": ms, "ms_per_iteration": per,
                      "algorithmic_bytes_per_iteration": bytes_it, "achieved_GBps": bytes_it / (per * 1e-3) / 1e9, "roofline_frac": bytes_it / (per * 1e-3) / 8e12,
                      "nnz_iter_per_s": nnz / (per * 1e-3)}))


if __name__ == "__main__":
    main()
