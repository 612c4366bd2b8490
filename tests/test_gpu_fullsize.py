"""Full-size instances of the BASELINE configurations on ONE GPU (pytest -m gpu), checked through size-independent
sampling: row blocks of one fused step are regenerated on the CPU from the counter-based generator and must equal
the oracle bit for bit.

  * the HEADLINE instances of bench.py: n = 10^7, 16/row, uniform columns (SURVEY §8(d)'s S-DD) and band half-width 4096;
  * one interior rank's slice of config 5 (n = 8*10^7 over 8 GPUs): rows [3*10^7, 4*10^7), n_cols = 8*10^7 — the layout,
    the kernels' window / panel arithmetic and the 640 MB gathered vector at the size an 8-GPU run gives every rank
    (row-chunk semantics of simd_ops.rs:219-238: a chunk's rows against the whole vector).

ms/step of every instance is appended to gpurun_out/fullsize_steps.jsonl when that directory exists (measurement log)."""
import ctypes as C
import json
from pathlib import Path

import numpy as np
import pytest

from sublinear_time_solver_amd import _lib as L
from sublinear_time_solver_amd import generators as G
from oracle import oracle as O

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def _fused_step_sampled(n_global, lo, hi, k, seed, w, sample_starts, label, steps=10):
    import torch
    lib = L.load()
    dev = torch.device("cuda", 0)
    rows = hi - lo
    rp = torch.empty(rows + 1, dtype=torch.int32, device=dev)
    ci = torch.empty(rows * k, dtype=torch.int32, device=dev)
    va = torch.empty(rows * k, dtype=torch.float64, device=dev)
    b = torch.empty(rows, dtype=torch.float64, device=dev)
    L.check(lib.sl_synth_sdd_device(n_global, k, seed, w, lo, hi, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), b.data_ptr()))
    h = C.c_void_p()
    L.check(lib.sl_matrix_create_csr(rows, n_global, rows * k, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), L.SL_MEM_DEVICE, lo, 0, C.byref(h)))
    del rp, ci, va
    torch.cuda.empty_cache()
    try:
        info = L.MatrixInfo()
        L.check(lib.sl_matrix_get_info(h, C.byref(info)))
        dinv = torch.empty(rows, dtype=torch.float64, device=dev)
        L.check(lib.sl_matrix_diagonal_inverse(h, dinv.data_ptr(), L.SL_MEM_DEVICE))
        # the gathered vector over ALL n_global columns (closed form of the index: t0 = D^-1 b, as bench.py starts)
        idx = torch.arange(n_global, device=dev, dtype=torch.float64)
        t0 = (1.0 + 0.001 * torch.remainder(idx, 1000.0)) * (1.0 / (10.0 + 0.01 * torch.remainder(idx, 1000.0)))
        del idx
        x = t0[lo:hi].clone()
        t1 = torch.zeros(rows, dtype=torch.float64, device=dev)
        norm2 = torch.zeros(2, dtype=torch.float64, device=dev)
        L.check(lib.sl_neumann_step(h, dinv.data_ptr(), t0.data_ptr(), t1.data_ptr(), x.data_ptr(), norm2.data_ptr(), 0))
        L.check(lib.sl_synchronize())
        t0h = t0.cpu().numpy()
        t1h, xh, dh = t1.cpu().numpy(), x.cpu().numpy(), dinv.cpu().numpy()
        for s0 in sample_starts:
            a, e = s0, s0 + 4096
            rrp, rci, rva, _ = G.sdd_rows(n_global, k, seed, w, lo + a, lo + e)
            y = O.spmv(rrp, rci, rva, t0h)                                      # rows of A t0, the oracle's order
            d = np.array([rva[i * k:(i + 1) * k][rci[i * k:(i + 1) * k] == lo + a + i][0] for i in range(e - a)])
            assert (bits(dh[a:e]) == bits(1.0 / d)).all(), f"{label}: dinv rows {a}..{e}"
            tn = t0h[lo + a:lo + e] - y * (1.0 / d)
            bad = np.nonzero(bits(t1h[a:e]) != bits(tn))[0]
            assert bad.size == 0, f"{label}: fused step rows {lo + a}..{lo + e}: {bad.size} differ, first {bad[:4]}"
            assert (bits(xh[a:e]) == bits(t0h[lo + a:lo + e] + tn)).all(), f"{label}: x rows {a}..{e}"
        assert abs(float(norm2[0]) - float(np.dot(t1h, t1h))) <= 1e-10 * float(norm2[0])
        # timing of the same launch (the gathered vector keeps its full length; only the local rows are rewritten)
        ta = t0.clone()
        tb = torch.zeros(rows, dtype=torch.float64, device=dev)
        ms = C.c_float(0)
        # run_steps ping-pongs t_a / t_b of n_rows entries; for a row slice the gathered vector is t_a itself, so time
        # single steps against the full-length vector instead
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        stream = torch.cuda.current_stream(dev)
        L.check(lib.sl_set_stream(C.c_void_p(stream.cuda_stream)))
        for _ in range(3):
            L.check(lib.sl_neumann_step(h, dinv.data_ptr(), ta.data_ptr(), tb.data_ptr(), x.data_ptr(), norm2.data_ptr(), 0))
        e0.record(stream)
        for _ in range(steps):
            L.check(lib.sl_neumann_step(h, dinv.data_ptr(), ta.data_ptr(), tb.data_ptr(), x.data_ptr(), norm2.data_ptr(), 0))
        e1.record(stream)
        torch.cuda.synchronize(dev)
        L.check(lib.sl_set_stream(None))
        per = e0.elapsed_time(e1) / steps
        rec = {"instance": label, "n_global": n_global, "rows": rows, "row_offset": lo, "k": k, "half_bandwidth": w,
               "ms_per_step": per, "nnz_iter_per_s": rows * k / (per * 1e-3),
               "roofline_frac": (12 * rows * k + 4 * (rows + 1) + 40 * rows) / (per * 1e-3) / 8e12,
               "column_panels": int(info.column_panels), "bandwidth_measured": int(info.bandwidth)}
        print(json.dumps(rec))
        out = ROOT / "gpurun_out"
        if out.is_dir():
            with open(out / "fullsize_steps.jsonl", "a") as f:
                f.write(json.dumps(rec) + "\n")
        return rec
    finally:
        lib.sl_matrix_destroy(h)
        torch.cuda.empty_cache()


@pytest.mark.parametrize("w", [4096, 0, 32768], ids=["band4096", "uniform", "band32768"])
def test_headline_instances_sampled_rows(gpu, w):
    """bench.py's two headline inputs at their own size: n = 10^7, 16/row, seed 1 (the kernel instance the bench times)."""
    n = 10_000_000
    rec = _fused_step_sampled(n, 0, n, 16, 1, w, (0, 4_999_937, n - 4096), f"C3 w={w}")
    # which layout the instance runs on: row slices behind the LDS window, paced column panels for uniform columns and — with block-local
    # rows and narrow panels — for the band too wide for the window (bench.py's secondary line)
    assert rec["column_panels"] == {4096: 0, 0: 2, 32768: 3}[w]


@pytest.mark.parametrize("w", [4096, 0, 2_500_000], ids=["band4096", "uniform", "locality_n_over_4P"])
def test_c5_interior_rank_slice(gpu, w):
    """Rank 3 of 8 of config 5: 10^7 rows at row offset 3*10^7 of the n = 8*10^7 system, n_cols = 8*10^7.  w = 2.5 * 10^6 is SURVEY
    8(e)'s halo variant of C5 (locality w = n / (4 P)): gathers spread over 40 MB of the vector — far beyond any cache — but a halo of
    2 w entries per neighbour instead of every row of every rank."""
    n, r = 80_000_000, 3
    lo, hi = r * 10_000_000, (r + 1) * 10_000_000
    _fused_step_sampled(n, lo, hi, 16, 1, w, (0, 5_000_011, hi - lo - 4096), f"C5 rank {r}/8 w={w}")
