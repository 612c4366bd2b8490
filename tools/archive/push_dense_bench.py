#!/usr/bin/env python3
"""ms per DENSE push round (row kernel, PUSH epilogue) on S-DD(n, k, w) built in HBM: theta small, dense switch ~0."""
import argparse
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=4_000_000)
    ap.add_argument("--k", type=int, default=16)
    ap.add_argument("--bandwidth", type=int, default=4096)
    ap.add_argument("--order", type=int, default=0)
    a = ap.parse_args()
    import torch
    from sublinear_time_solver_amd import _lib as L
    lib = L.load()
    n, k = a.rows, a.k
    dev = torch.device("cuda", 0)
    rp = torch.empty(n + 1, dtype=torch.int32, device=dev)
    ci = torch.empty(n * k, dtype=torch.int32, device=dev)
    va = torch.empty(n * k, dtype=torch.float64, device=dev)
    b = torch.empty(n, dtype=torch.float64, device=dev)
    L.check(lib.sl_synth_sdd_device(n, k, 1, a.bandwidth, 0, n, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), b.data_ptr()))
    h = C.c_void_p()
    L.check(lib.sl_matrix_create_csr(n, n, n * k, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), L.SL_MEM_DEVICE, 0, L.SL_MATRIX_WITH_TRANSPOSE, C.byref(h)))
    o = L.PushOptions()
    lib.sl_push_options_default(C.byref(o))
    o.theta, o.max_rounds, o.order, o.mem, o.dense_switch = 1e-9, 1000, a.order, L.SL_MEM_DEVICE, 1e-12
    for rep in range(2):
        x = torch.zeros(n, dtype=torch.float64, device=dev)
        res = L.PushResult()
        L.check(lib.sl_push_solve(h, b.data_ptr(), C.byref(o), x.data_ptr(), None, None, 0, None, C.byref(res)))
    print(f"k={k} w={a.bandwidth} order={a.order}: {res.rounds} rounds ({res.dense_rounds} dense), {res.device_time_ms / res.rounds:.4f} ms/round")


if __name__ == "__main__":
    main()
