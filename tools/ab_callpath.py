#!/usr/bin/env python3
"""The same instance, the same binary, two call paths in ONE process, interleaved: K fused steps through sl_neumann_run_steps (ping-pong
term buffers, HIP events inside the library) against K single sl_neumann_step calls (one gathered vector read every time, a separate
output, torch events around the loop) — VERDICT r02 item 7: the 16 % spread of the w = 32768 figure between bench.py's sweep and
tests/test_gpu_fullsize.py."""
import argparse
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--k", type=int, default=16)
    ap.add_argument("--bandwidth", type=int, default=32768)
    ap.add_argument("--steps", type=int, default=30)
    a = ap.parse_args()
    import torch
    from sublinear_time_solver_amd import _lib as L
    lib = L.load()
    dev = torch.device("cuda", 0)
    n, k = a.rows, a.k
    rp = torch.empty(n + 1, dtype=torch.int32, device=dev); ci = torch.empty(n * k, dtype=torch.int32, device=dev)
    va = torch.empty(n * k, dtype=torch.float64, device=dev); b = torch.empty(n, dtype=torch.float64, device=dev)
    L.check(lib.sl_synth_sdd_device(n, k, 1, a.bandwidth, 0, n, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), b.data_ptr()))
    h = C.c_void_p()
    L.check(lib.sl_matrix_create_csr(n, n, n * k, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), L.SL_MEM_DEVICE, 0, 0, C.byref(h)))
    del rp, ci, va
    torch.cuda.empty_cache()
    dinv = torch.empty(n, dtype=torch.float64, device=dev)
    L.check(lib.sl_matrix_diagonal_inverse(h, dinv.data_ptr(), L.SL_MEM_DEVICE))
    t0 = b * dinv
    stream = torch.cuda.current_stream(dev)
    L.check(lib.sl_set_stream(C.c_void_p(stream.cuda_stream)))
    nrm = torch.zeros(2, dtype=torch.float64, device=dev)
    for rep in range(4):
        # path A: run_steps (what bench.py's sweep times)
        ta, tb, x = t0.clone(), torch.empty_like(t0), t0.clone()
        ms = C.c_float(0)
        L.check(lib.sl_neumann_run_steps(h, dinv.data_ptr(), ta.data_ptr(), tb.data_ptr(), x.data_ptr(), nrm.data_ptr(), 0, 3, C.byref(ms)))
        L.check(lib.sl_neumann_run_steps(h, dinv.data_ptr(), tb.data_ptr(), ta.data_ptr(), x.data_ptr(), nrm.data_ptr(), 0, a.steps, C.byref(ms)))
        run_steps = ms.value / a.steps
        # path B: single steps, ping-pong by hand
        ta, tb, x = t0.clone(), torch.empty_like(t0), t0.clone()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            L.check(lib.sl_neumann_step(h, dinv.data_ptr(), ta.data_ptr(), tb.data_ptr(), x.data_ptr(), nrm.data_ptr(), 0)); ta, tb = tb, ta
        e0.record(stream)
        for _ in range(a.steps):
            L.check(lib.sl_neumann_step(h, dinv.data_ptr(), ta.data_ptr(), tb.data_ptr(), x.data_ptr(), nrm.data_ptr(), 0)); ta, tb = tb, ta
        e1.record(stream); e1.synchronize()
        single_pp = e0.elapsed_time(e1) / a.steps
        # path C: single steps the way test_gpu_fullsize.py times them: always the SAME input vector, a separate output, x growing
        ta, tb, x = t0.clone(), torch.zeros_like(t0), t0.clone()
        for _ in range(3):
            L.check(lib.sl_neumann_step(h, dinv.data_ptr(), ta.data_ptr(), tb.data_ptr(), x.data_ptr(), nrm.data_ptr(), 0))
        e0.record(stream)
        for _ in range(a.steps):
            L.check(lib.sl_neumann_step(h, dinv.data_ptr(), ta.data_ptr(), tb.data_ptr(), x.data_ptr(), nrm.data_ptr(), 0))
        e1.record(stream); e1.synchronize()
        single_same = e0.elapsed_time(e1) / a.steps
        print(f"w={a.bandwidth} rep {rep}: run_steps {run_steps:.4f} ms   single steps ping-pong {single_pp:.4f} ms   single steps, same input every time {single_same:.4f} ms", flush=True)
    lib.sl_matrix_destroy(h)


if __name__ == "__main__":
    main()
