"""Multi-pass window kernel (sl_mpass_kernel): bands too wide for one LDS window (half-width > ~9.5 K) with rows of at most 16
entries — the window is staged segment by segment while the rows' matrix bytes wait in registers.  Every epilogue bit for bit
against the oracle: uniform width 16 and 8, ragged rows, a 7-point stencil whose window has three occupied clusters, a row slice.

The kernel is an OPT-IN (SL_MPASS=1; measured slower than the general kernel on the wide bands it was written for);
the library reads its knobs once per process, so this file runs its checks in a child pytest with the knob set — and the very
same checks in-process through the default (general) kernel."""
import os
import subprocess
import sys

import numpy as np
import pytest

import sublinear_time_solver_amd as S
from sublinear_time_solver_amd import generators as G
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def _check_all(rp, ci, va, b, n, tol=1e-10):
    m = S.SparseMatrix.from_csr(rp, ci, va, n, n, with_transpose=True)
    info = m.info()
    # the kernel's territory; in the child run with SL_PW_BAND set the same matrices carry the paced layout with block-local rows
    assert info.column_panels == (3 if int(os.environ.get("SL_PW_BAND", "0") or 0) > 0 else 0)
    assert info.bandwidth > 9500 and info.max_row_nnz <= 16 and info.n_long_rows == 0
    x = np.cos(np.arange(n) * 0.37) + 0.1
    assert (bits(m.multiply_vector(x)) == bits(O.spmv(rp, ci, va, x))).all()
    o = O.neumann_solve(rp, ci, va, b, tolerance=tol)
    g = S.NeumannSolver().solve(m, b, S.SolverOptions(tolerance=tol))
    assert g.converged and g.iterations == o["iterations"] and (bits(g.solution) == bits(o["x"])).all()
    np.testing.assert_allclose(g.term_norms, o["term_norms"], rtol=1e-10)
    assert abs(g.residual_norm - o["residual_norm"]) <= 1e-10 * max(1.0, o["residual_norm"])
    bs = b * (np.arange(n) % 4 == 0)
    q = O.push_sync_solve(rp, ci, va, bs, theta=1e-8, log_cap=1 << 22)
    p = S.PushSolver(theta=1e-8, dense_switch=1e-9).solve(m, bs, log_frontier=1 << 22)       # dense rounds: the PUSH epilogue
    assert p["converged"] and p["rounds"] == q["rounds"] and p["dense_rounds"] > 0
    assert (p["frontier_log"] == q["frontier_log"]).all() and (bits(p["solution"]) == bits(q["x"])).all() and (bits(p["residual"]) == bits(q["r"])).all()


@pytest.mark.parametrize("n,k,w", [(60_000, 16, 12_000), (90_001, 8, 30_000), (70_000, 12, 10_000)])
def test_wide_bands_uniform_rows(gpu, n, k, w):
    rp, ci, va, b = G.sdd_rows(n, k, seed=5, half_bandwidth=w)
    _check_all(rp, ci, va, b, n)


def test_ragged_rows_in_a_wide_band(gpu):
    rng = np.random.default_rng(8)
    n, w = 50_000, 20_000
    tr, tc, tv = [], [], []
    for i in range(n):
        m = int(rng.integers(1, 16))                                     # 1..15 off-diagonal entries + the diagonal
        lo, hi = max(0, i - w), min(n, i + w + 1)
        cols = np.unique(rng.integers(lo, hi, size=m))
        cols = cols[cols != i]
        vals = rng.uniform(-1.0, 1.0, size=cols.size)
        tr += [i] * (cols.size + 1); tc += cols.tolist() + [i]; tv += vals.tolist() + [2.0 * np.abs(vals).sum() + 1.0]
    rp, ci, va = O.csr_from_triplets(tr, tc, tv, n, n)
    _check_all(rp, ci, va, 1.0 + (np.arange(n) % 13) * 0.25, n)


def test_duplicates_and_bare_rows_in_a_wide_band(gpu):
    """the same column stored twice in a row (kept, added twice in stored order: a run of two in one panel) and rows that hold only
    their diagonal, scattered through a wide band"""
    rng = np.random.default_rng(21)
    n, w = 64_000, 15_000
    tr, tc, tv = [], [], []
    for i in range(n):
        m = 0 if i % 11 == 0 else int(rng.integers(2, 13))
        lo, hi = max(0, i - w), min(n, i + w + 1)
        cols = np.unique(rng.integers(lo, hi, size=m)) if m else np.zeros(0, dtype=np.int64)
        cols = cols[cols != i]
        if i % 7 == 0 and cols.size > 2:
            cols = np.sort(np.concatenate([cols, cols[:2]]))                # two duplicated columns
        vals = rng.uniform(-1.0, 1.0, size=cols.size)
        order = np.argsort(np.concatenate([cols, [i]]), kind="stable")
        allc = np.concatenate([cols, [i]])[order]
        allv = np.concatenate([vals, [2.0 * np.abs(vals).sum() + 1.0]])[order]
        tr += [i] * allc.size; tc += allc.tolist(); tv += allv.tolist()
    rp = np.zeros(n + 1, dtype=np.uint32)
    rp[1:] = np.cumsum(np.bincount(np.array(tr), minlength=n))
    _check_all(rp, np.array(tc, dtype=np.uint32), np.array(tv), 1.0 + (np.arange(n) % 17) * 0.125, n)


def test_seven_point_stencil_three_clusters(gpu):
    nx, ny, nz = 110, 110, 8                                               # bandwidth nx * ny = 12100, rows of 4..7 entries
    n = nx * ny * nz
    i = np.arange(n)
    x, y, z = i % nx, (i // nx) % ny, i // (nx * ny)
    offs = [(-nx * ny, z > 0, -1.00), (-nx, y > 0, -0.90), (-1, x > 0, -0.80), (0, np.ones(n, bool), 12.0), (1, x < nx - 1, -1.10),
            (nx, y < ny - 1, -0.95), (nx * ny, z < nz - 1, -1.05)]
    mask = np.stack([m for _, m, _ in offs], axis=1)
    cols = np.stack([i + o for o, _, _ in offs], axis=1)
    vals = np.tile(np.array([v for _, _, v in offs]), (n, 1))
    rp = np.zeros(n + 1, dtype=np.uint32)
    rp[1:] = np.cumsum(mask.sum(axis=1))
    _check_all(rp, cols[mask].astype(np.uint32), vals[mask], 1.0 + 0.01 * (i % 97), n, tol=1e-8)


def test_row_slice_of_a_wide_band(gpu):
    n, k, w, lo, hi = 120_000, 16, 14_000, 41_111, 99_999
    rp, ci, va, b = G.sdd_rows(n, k, seed=9, half_bandwidth=w)
    prp = (rp[lo:hi + 1].astype(np.int64) - int(rp[lo])).astype(np.uint32)
    m = S.SparseMatrix.from_csr(prp, ci[rp[lo]:rp[hi]], va[rp[lo]:rp[hi]], hi - lo, n, row_offset=lo)
    x = np.sin(np.arange(n) * 0.05) + 0.3
    assert (bits(m.multiply_vector(x)) == bits(O.spmv(rp, ci, va, x)[lo:hi])).all()


def _inside_child():
    return os.environ.get("SL_MPASS") == "1" or int(os.environ.get("SL_PW_BAND", "0") or 0) > 0


@pytest.mark.parametrize("bits_", ["9", "11"])
def test_the_same_checks_through_the_wide_band_panel_layout(gpu, bits_):
    """wide bands at full size run on the paced column-panel layout with block-local rows and panels of 2^9 / 2^10 columns (selected
    from 1.5 * 10^6 rows on); here it is forced on these small systems (SL_PW_BAND = log2 of the panel width, a pretended 2-CU device:
    several rounds of tiles) and must give the same bits: SpMV, Neumann solve, residual, dense push rounds, ragged rows, the stencil
    whose window has three occupied clusters, a row slice"""
    if _inside_child():
        pytest.skip("already inside a child run")
    env = dict(os.environ, SL_PW_BAND=bits_, SL_PW_FORCE="1", SL_PW_CUS="2")
    r = subprocess.run([sys.executable, "-m", "pytest", __file__, "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, timeout=1200, env=env)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_the_same_checks_through_the_multi_pass_kernel(gpu):
    if _inside_child():
        pytest.skip("already inside the child run")
    r = subprocess.run([sys.executable, "-m", "pytest", __file__, "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, timeout=1200, env=dict(os.environ, SL_MPASS="1"))
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
