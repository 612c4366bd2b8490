"""The LDS band kernel (sl_band_kernel, uniform-width pipelined path) over every shape of a wave's run of slices.

The pipelined loop keeps two slices of matrix bytes in flight: a steady-state body whose loads are unconditional (both successors of
the current slice exist) and a tail of one or two slices.  How many slices a wave owns follows from the block geometry (waves per
block, slices per wave) and from where the matrix ends — so the geometry knobs (read once per process: a child process per setting)
are swept over matrices whose last block is cut at every position: one slice, two, odd counts, a last slice with fewer than 64
rows, a matrix smaller than one block.  Every result must have the bits of the sequential reference loop (sparse.rs:187-203 through the
fused Neumann step, neumann.rs:252-299, and the dense push round)."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

PROG = r"""
import sys, numpy as np
from oracle import oracle as O
import sublinear_time_solver_amd as S

def banded(n, k, hb, seed):
    rng = np.random.default_rng(seed)
    rp = np.arange(n + 1, dtype=np.uint32) * k
    ci = np.zeros(n * k, dtype=np.uint32); va = np.zeros(n * k)
    for i in range(n):
        lo, hi = max(0, i - hb), min(n - 1, i + hb)
        cand = np.setdiff1d(np.arange(lo, hi + 1), [i])
        c = np.sort(np.append(rng.choice(cand, k - 1, replace=False), i))
        v = rng.uniform(-1, 1, k); v[c == i] = 2.0 * np.abs(v).sum() + 1
        ci[i * k:(i + 1) * k] = c; va[i * k:(i + 1) * k] = v
    return rp, ci, va, rng.standard_normal(n)

cases = 0
for k in (8, 16):
    for hb in (30, 3000):
        for n in (100, 64 * 4 * 3 + 1, 64 * 8 * 3 + 65, 4100):
            if hb >= n:
                continue
            rp, ci, va, b = banded(n, k, hb, n + k + hb)
            x = np.cos(np.arange(n) * 0.37)
            m = S.SparseMatrix.from_csr(rp, ci, va, n, n, with_transpose=True)
            assert m.info().bandwidth <= hb
            assert (m.multiply_vector(x).view(np.uint64) == O.spmv(rp, ci, va, x).view(np.uint64)).all(), (k, hb, n)
            g = S.NeumannSolver().solve(m, b, S.SolverOptions(tolerance=1e-9, max_iterations=100))
            o = O.neumann_solve(rp, ci, va, b, tolerance=1e-9, max_iterations=100)
            assert g.iterations == o["iterations"] and (g.solution.view(np.uint64) == o["x"].view(np.uint64)).all(), (k, hb, n)
            # the thresholded push with every round dense (dense_switch below any frontier): the same kernel with the push epilogue, whose
            # threshold / column operands travel with the slice's other vectors
            if n <= 1000 or (k == 8 and hb == 30):
                g = S.PushSolver(theta=1e-7, dense_switch=1e-9).solve(m, b)
                o = O.push_sync_solve(rp, ci, va, b, theta=1e-7)
                assert g["rounds"] == o["rounds"] and g["pushes"] == o["pushes"] and g["dense_rounds"] > 0, (k, hb, n)
                assert (g["solution"].view(np.uint64) == o["x"].view(np.uint64)).all() and (g["residual"].view(np.uint64) == o["r"].view(np.uint64)).all(), (k, hb, n)
                thr = 1e-7 * (1.0 + (np.arange(n) % 5))                   # per-row thresholds (the degree-scaled rule's operand)
                g = S.PushSolver(theta=1e-7, dense_switch=1e-9, theta_rows=thr).solve(m, b)
                o = O.push_sync_solve(rp, ci, va, b, theta=1e-7, theta_rows=thr)
                assert g["rounds"] == o["rounds"] and g["pushes"] == o["pushes"], (k, hb, n, "theta_rows")
                assert (g["solution"].view(np.uint64) == o["x"].view(np.uint64)).all() and (g["residual"].view(np.uint64) == o["r"].view(np.uint64)).all(), (k, hb, n)
            cases += 1
print("band geometry ok", cases)
"""


@pytest.mark.parametrize("knobs", [{}, {"SL_BAND_SPW": "1"}, {"SL_BAND_SPW": "2"}, {"SL_BAND_SPW": "5"}, {"SL_BAND_NW": "16"},
                                   {"SL_BAND_PIPE": "0"}, {"SL_BAND_C16": "0", "SL_BAND_SPW": "3"}],
                         ids=lambda k: ",".join(f"{a}={b}" for a, b in k.items()) or "default")
def test_band_kernel_every_slice_run_shape(gpu, knobs):
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, **knobs)
    r = subprocess.run([sys.executable, "-c", PROG], cwd=root, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "band geometry ok 10" in r.stdout, r.stdout[-500:] + r.stderr[-2000:]
