#!/usr/bin/env python3
"""The fused step on a STRUCTURED system: 7-point stencil on an nx x ny x nz grid in natural order (offsets +-1, +-nx,
+-nx*ny; rows at the faces are shorter), asymmetric weights, strictly row dominant.  Gathers at a fixed offset are
coalesced across a wavefront, so the general kernel streams even though the bandwidth (nx*ny) is far beyond the LDS window."""
import argparse
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=215)
    ap.add_argument("--ny", type=int, default=215)
    ap.add_argument("--nz", type=int, default=215)
    ap.add_argument("--steps", type=int, default=40)
    a = ap.parse_args()
    import torch
    from sublinear_time_solver_amd import _lib as L
    lib = L.load()
    dev = torch.device("cuda", 0)
    nx, ny, nz = a.nx, a.ny, a.nz
    n = nx * ny * nz
    i = torch.arange(n, device=dev, dtype=torch.int64)
    x, y, z = i % nx, (i // nx) % ny, i // (nx * ny)
    # neighbours in ascending column order: -nx*ny, -nx, -1, (diag), +1, +nx, +nx*ny ; weights differ per direction (asymmetric)
    offs = [(-nx * ny, z > 0, -1.00), (-nx, y > 0, -0.90), (-1, x > 0, -0.80), (0, torch.ones_like(x, dtype=torch.bool), 6.0),
            (1, x < nx - 1, -1.10), (nx, y < ny - 1, -0.95), (nx * ny, z < nz - 1, -1.05)]
    mask = torch.stack([m for _, m, _ in offs], dim=1)                         # n x 7
    cols = torch.stack([i + o for o, _, _ in offs], dim=1)
    vals = torch.tensor([v for _, _, v in offs], dtype=torch.float64, device=dev).repeat(n, 1)
    counts = mask.sum(dim=1)
    rp = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    rp[1:] = torch.cumsum(counts, 0)
    ci = cols[mask].to(torch.int32)
    va = vals[mask].contiguous()
    nnz = int(rp[-1])
    rp32 = rp.to(torch.int32)
    h = C.c_void_p()
    L.check(lib.sl_matrix_create_csr(n, n, nnz, rp32.data_ptr(), ci.data_ptr(), va.data_ptr(), L.SL_MEM_DEVICE, 0, 0, C.byref(h)))
    info = L.MatrixInfo()
    L.check(lib.sl_matrix_get_info(h, C.byref(info)))
    dd = C.c_int(0)
    L.check(lib.sl_matrix_is_diagonally_dominant(h, C.byref(dd)))
    dinv = torch.empty(n, dtype=torch.float64, device=dev)
    L.check(lib.sl_matrix_diagonal_inverse(h, dinv.data_ptr(), L.SL_MEM_DEVICE))
    b = torch.ones(n, dtype=torch.float64, device=dev)
    ta = b * dinv
    tb = torch.empty_like(ta)
    xv = ta.clone()
    nrm = torch.zeros(2, dtype=torch.float64, device=dev)
    ms = C.c_float(0)
    L.check(lib.sl_neumann_run_steps(h, dinv.data_ptr(), ta.data_ptr(), tb.data_ptr(), xv.data_ptr(), nrm.data_ptr(), 0, 4, C.byref(ms)))
    L.check(lib.sl_neumann_run_steps(h, dinv.data_ptr(), ta.data_ptr(), tb.data_ptr(), xv.data_ptr(), nrm.data_ptr(), 0, a.steps, C.byref(ms)))
    per = ms.value / a.steps
    alg = 12 * nnz + 4 * (n + 1) + 40 * n
    print(f"7-point stencil {nx}x{ny}x{nz}: n={n} nnz={nnz} ({nnz / n:.2f}/row, padded {info.padded_nnz / nnz:.2f}x) bandwidth={info.bandwidth} dd={dd.value} "
          f"{per:.4f} ms/step -> {nnz / (per * 1e-3):.3e} nnz*iter/s, {alg / (per * 1e-3) / 1e9:.0f} GB/s algorithmic = {alg / (per * 1e-3) / 8e12:.3f} of 8 TB/s")
    lib.sl_matrix_destroy(h)


if __name__ == "__main__":
    main()
