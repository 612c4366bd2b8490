"""bench.py contract on the GPU: JSON line fields, and the boundary / interior split used for the multi-GPU
overlap must not change the iteration (same last term norm as the single-launch step)."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _bench(*extra):
    r = subprocess.run([sys.executable, "bench.py", "--n", "300000", "--k", "16", "--bandwidth", "512", "--steps", "7", "--warmup", "0",
                        "--no-cpu-baseline", "--no-sweep", *extra], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def test_bench_line_contract_and_split_equivalence(gpu):
    a = _bench()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in a
    assert a["n_gpus"] == 1 and a["steps"] == 7 and a["dtype"] == "f64" and a["scaling"] == "weak" and a["vs_baseline"] is None
    assert set(a["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and a["roofline"]["bound"] == "hbm"
    assert abs(a["roofline"]["frac"] - a["roofline"]["achieved"] / 8000.0) < 1e-12 and "workload" in a["config"]
    b = _bench("--force-split")
    assert b["config"]["exchange"] == "halo+overlap"
    na, nb = a["config"]["last_term_norm"], b["config"]["last_term_norm"]
    assert abs(na - nb) <= 1e-12 * na and na > 0
