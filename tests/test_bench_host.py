"""Host-side pieces of bench.py that need no GPU: the CPU checker of the parity gate (it must accept what the reference's arithmetic
produces and must notice one flipped bit), the generator for scattered rows behind it, and the launcher's time budget."""
import importlib.util
import sys
from pathlib import Path

import numpy as np

from oracle import oracle as O
from sublinear_time_solver_amd import generators as G

ROOT = Path(__file__).resolve().parent.parent


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", ROOT / "bench.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_scattered_rows_equal_row_ranges():
    n, k, seed = 200_000, 16, 3
    for w in (0, 700, 50_000):
        want = np.array([5, 199_999, 0, 77_777, 5], dtype=np.uint64)
        rp, ci, va, b = G.sdd_rows_at(n, k, seed, w, want)
        for j, r in enumerate(want.tolist()):
            _, c1, v1, b1 = G.sdd_rows(n, k, seed, w, r, r + 1)
            assert (ci[j * k:(j + 1) * k] == c1).all() and (va[j * k:(j + 1) * k].view(np.uint64) == v1.view(np.uint64)).all() and b[j] == b1[0]


def _two_steps_by_the_oracle(n_global, k, seed, w, lo, hi, order):
    """what a rank holds after two fused steps from t0 = D^-1 b, x0 = t0 — computed over the WHOLE system by the oracle's step loop"""
    rp, ci, va, b = G.sdd_rows(n_global, k, seed, w)
    dinv = 1.0 / (10.0 + 0.01 * (np.arange(n_global) % 1000))
    t = b * dinv
    x = t.copy()
    O.neumann_steps(rp, ci, va, dinv, t, x, 2, order)
    return t[lo:hi], x[lo:hi]


def test_parity_gate_checker_accepts_the_reference_arithmetic_and_sees_one_flipped_bit(tmp_path):
    B = _bench()
    n_global, k, seed = 60_000, 16, 1
    for w, lo, hi, order in ((0, 0, 60_000, 0), (7_500, 30_000, 45_000, 0), (512, 15_000, 30_000, 1)):
        t, x = _two_steps_by_the_oracle(n_global, k, seed, w, lo, hi, order)
        starts = B.gate_block_starts(hi - lo)
        assert starts[0] == 0 and starts[-1] == hi - lo - B.GATE_BLOCK and len(starts) == 3
        term = np.stack([t[s:s + B.GATE_BLOCK] for s in starts])
        xs = np.stack([x[s:s + B.GATE_BLOCK] for s in starts])
        f = tmp_path / f"gate_{w}.npz"
        np.savez(f, n_global=n_global, k=k, seed=seed, w=w, lo=lo, order=order, steps=2, starts=np.asarray(starts, dtype=np.int64), term=term, x=xs)
        ok = B.cpu_parity_gate(str(f))
        assert ok["bitwise_equal"] and ok["rows_checked"] == 3 * B.GATE_BLOCK and ok["max_rel_err"] == 0.0, ok
        term.view(np.uint64)[1, 17] ^= 1                                  # one ulp in one row of one block
        np.savez(f, n_global=n_global, k=k, seed=seed, w=w, lo=lo, order=order, steps=2, starts=np.asarray(starts, dtype=np.int64), term=term, x=xs)
        bad = B.cpu_parity_gate(str(f))
        assert not bad["bitwise_equal"] and bad["values_differing"] == 1 and 0 < bad["max_rel_err"] < 1e-15


def test_launcher_time_budget_leaves_every_transport_its_turn():
    """three attempts at the default limits fit the driver's 1800 s with room to spare"""
    B = _bench()
    import argparse
    src = (ROOT / "bench.py").read_text()
    assert '"SL_BENCH_ATTEMPT_TIMEOUT", "420"' in src and '"SL_BENCH_TOTAL_TIMEOUT", "1440"' in src
    assert 3 * 420 <= 1440 < 1500


def test_recorded_traffic_is_reported_only_for_the_kernel_it_was_measured_on():
    """roofline.traffic comes from committed rocprofv3 counter passes; a record names the sl_kernels.hip it was taken on and is withheld
    (traffic = null, the reason in traffic_source) once that file has changed (the round-3 record was 'stale by construction')"""
    import hashlib
    import importlib.util
    import json
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    spec = importlib.util.spec_from_file_location("bench_mod", root / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    now = hashlib.sha256((root / "sublinear_time_solver_amd" / "csrc" / "sl_kernels.hip").read_bytes()).hexdigest()[:16]
    recs = json.loads((root / "profiles" / "pmc_traffic.json").read_text())["records"]
    assert all("kernel_source_sha16" in r and "commit" in r for r in recs.values())
    for r in recs.values():
        got, why = bench.recorded_traffic(r["n"], r["k"], r["bandwidth"])
        same_kernel = any(q["kernel_source_sha16"] == now for q in recs.values() if (q["n"], q["k"], q["bandwidth"]) == (r["n"], r["k"], r["bandwidth"]))
        if same_kernel:
            assert got and "this very kernel source" in why
        else:
            assert got is None and "re-profile" in why and now in why
    assert bench.recorded_traffic(123, 4, 5) == (None, None)          # no record at all: nothing to say
