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
    assert set(a["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "column_structure"} and a["roofline"]["bound"] == "hbm"
    assert abs(a["roofline"]["frac"] - a["roofline"]["achieved"] / 8000.0) < 1e-12 and "workload" in a["config"]
    # the parity gate ran before the timed region: three blocks of 4096 rows of the term and the solution after two steps, bit for bit
    g = a["parity_gate"]
    assert g["bitwise_equal"] is True and g["rows_checked"] == 3 * 4096 and g["max_rel_err"] == 0.0 and g["ranks_checked"] == 1
    b = _bench("--force-split")
    assert b["config"]["exchange"] == "halo+overlap"
    na, nb = a["config"]["last_term_norm"], b["config"]["last_term_norm"]
    assert abs(na - nb) <= 1e-12 * na and na > 0


def _bench_two_ranks(n_per_rank, w, *extra):
    """two ranks sharing the one GPU of the test box; exchanges staged through the host over gloo (SL_BENCH_BACKEND)"""
    import os
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, SL_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), "bench.py", "--gpus", "2", "--rows", str(n_per_rank), "--k", "16", "--bandwidth", str(w),
                        "--steps", "7", "--warmup", "0", "--no-cpu-baseline", "--no-sweep", *extra],
                       cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


@pytest.mark.parametrize("w,extra,exchange", [(512, ("--exchange", "p2p"), "halo+overlap"), (512, ("--exchange", "p2p", "--no-overlap"), "halo"),
                                              (0, ("--exchange", "p2p"), "allgather"),
                                              (512, ("--exchange", "allreduce"), "halo_allreduce+overlap"),
                                              (512, (), "abi"), (0, (), "abi")])       # the default: the library's own communicator (C ABI)
def test_bench_two_ranks_reproduce_the_single_rank_iteration(gpu, w, extra, exchange):
    """bench.py at world size 2 (row slices at a non-zero row offset, halo / all-gather exchange, boundary-first overlap,
    norm all-reduce) must run the same iteration as one rank owning all rows: same last term norm"""
    one = subprocess.run([sys.executable, "bench.py", "--n", "300000", "--k", "16", "--bandwidth", str(w), "--steps", "7", "--warmup", "0",
                          "--no-cpu-baseline", "--no-sweep"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    a = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    b = _bench_two_ranks(150000, w, *extra)
    assert b["n_ranks"] == 2 and b["n_gpus"] == 1 and b["config"]["n_global"] == 300000 and b["scaling"] == "weak"       # two ranks SHARING the box's GPU
    if exchange == "abi":
        assert b["parity_gate"]["bitwise_equal"] is True and b["parity_gate"]["ranks_checked"] == 2 and b["parity_gate"]["rows_checked"] == 2 * 3 * 4096
    assert b["config"]["exchange"].startswith("abi: sl_comm") if exchange == "abi" else b["config"]["exchange"] == exchange
    na, nb = a["config"]["last_term_norm"], b["config"]["last_term_norm"]
    assert na > 0 and abs(na - nb) <= 1e-12 * na
    assert abs(b["value"] - 300000 * 16 * 7 / (b["ms_per_step"] * 7e-3)) <= 1e-6 * b["value"]     # whole-job units / max-over-ranks time


def test_bench_starts_its_own_ranks(gpu):
    """exactly what the driver runs for N > 1 — `python bench.py --gpus 2 ...`, no launcher around it: the parent starts one process
    per rank (sharing the box's GPU here), the ranks meet in sl_comm, the exchange is verified against the owners' copies, rank 0's
    line comes out of the parent; and the same iteration as one rank owning all rows"""
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--rows", "300000", "--steps", "5"], cwd=ROOT, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    a = json.loads(lines[0])
    cfg = a["config"]
    assert a["n_ranks"] == 2 and a["n_gpus"] == len(set(cfg["devices"])) == 1 and "2 ranks on 1 device(s)" in cfg["workload"]
    assert a["steps"] == 5 and a["warmup"] == 5 and cfg["n_global"] == 600000 and a["scaling"] == "weak"
    assert cfg["n_ranks_joined"] == 2 and len(cfg["devices"]) == 2 and cfg["transport"] == "ipc" and cfg["exchange_verified"] is True
    assert a["parity_gate"]["bitwise_equal"] is True and a["parity_gate"]["ranks_equal"] == 2
    ref = a["scaling_reference"]                                      # the like-for-like one-GPU figures, measured in the same job by rank 0 alone
    assert ref["n1_ms_per_step"] > 0 and ref["slice_ms_per_step"] > 0
    assert cfg["launcher"]["self_launched_ranks"] == 2 and cfg["launcher"]["attempts"][-1]["ok"]
    # N > 1 defaults to config 5's locality-bounded form (columns within n_local / 4 of the row); the all-gather form and the narrow band beside it
    assert cfg["half_bandwidth"] == 75000 and "locality-bounded" in cfg["workload"]
    for key in ("halo_variant", "uniform_variant"):
        assert a[key]["exchange_verified"] is True and a[key]["value"] > 0 and a[key]["parity_gate"]["bitwise_equal"] is True
    assert a["uniform_variant"]["bytes_received_per_rank_per_step"] == 8 * 300000 and cfg["bytes_received_per_rank_per_step"] == 8 * 75000
    assert abs(a["value"] - 600000 * 16 * 5 / (a["ms_per_step"] * 5e-3)) <= 1e-6 * a["value"]
    one = subprocess.run([sys.executable, "bench.py", "--rows", "600000", "--steps", "5", "--bandwidth", "75000", "--no-cpu-baseline", "--no-sweep"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    b = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    assert b["n_gpus"] == 1 and b["config"]["n_ranks_joined"] == 1 and b["config"]["transport"] == "ipc"      # N = 1 runs the same host path
    na, nb = cfg["last_term_norm"], b["config"]["last_term_norm"]
    assert na > 0 and abs(na - nb) <= 1e-12 * na


def test_config5_at_its_own_size_eight_ranks_on_this_box(gpu):
    """BASELINE configs[4] as itself: `python bench.py --gpus 8 --rows 10000000` — 8 ranks x 10^7 rows of the n = 8 * 10^7 system, through the
    real rendezvous, IPC handles, tickets and pulls (the ranks share this box's one GPU: about 64 GB of its 288).  All eight join; the
    exchange verifies for the locality-bounded, the all-gather and the narrow-band form; and EVERY rank's sampled blocks (its first, middle
    and last 4096 rows after two steps — the edge blocks gather what the exchange moved) equal the CPU checker bit for bit."""
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--rows", "10000000", "--steps", "5", "--warmup", "2"], cwd=ROOT, capture_output=True, text=True, timeout=1700)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-4000:]
    a = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    cfg = a["config"]
    assert a["n_ranks"] == 8 and cfg["n_ranks_joined"] == 8 and len(cfg["devices"]) == 8 and a["n_gpus"] == len(set(cfg["devices"]))
    assert cfg["n_global"] == 80_000_000 and cfg["half_bandwidth"] == 2_500_000 and cfg["exchange_verified"] is True and a["value"] > 0
    for g in (a["parity_gate"], a["uniform_variant"]["parity_gate"], a["halo_variant"]["parity_gate"]):
        assert g["bitwise_equal"] is True and g["ranks_equal"] == 8 and g["rows_checked"] == 8 * 3 * 4096 and g["max_rel_err"] == 0.0
    assert a["uniform_variant"]["exchange_verified"] is True and a["halo_variant"]["exchange_verified"] is True
    assert a["uniform_variant"]["bytes_received_per_rank_per_step"] == 7 * 8 * 10_000_000
    ref = a["scaling_reference"]
    assert 0 < ref["n1_ms_per_step"] < 5 and 0 < ref["slice_ms_per_step"] < 5
    out = ROOT / "gpurun_out"
    if out.is_dir():
        (out / "bench_8ranks_1gpu.json").write_text(json.dumps(a, indent=1))


def test_bench_refuses_a_value_when_the_parity_gate_fails(gpu):
    """a step whose results differ from the checker's is not timed into a value (SL_BENCH_GATE_CORRUPT flips one bit of what the gate read back)"""
    import os
    r = subprocess.run([sys.executable, "bench.py", "--rows", "200000", "--steps", "3", "--warmup", "1", "--no-sweep", "--no-cpu-baseline"], cwd=ROOT,
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, SL_BENCH_GATE_CORRUPT="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    a = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert a["value"] is None and a["parity_gate"]["bitwise_equal"] is False and "parity gate failed" in a["error"]


def test_bench_launcher_falls_back_to_the_next_transport(gpu):
    """a rank that dies before the rendezvous: the attempt ends (bounded waits, then the parent's kill), the next one runs, the
    line records both"""
    import os
    env = dict(os.environ, SL_BENCH_TRANSPORTS="ipc,ipc", SL_BENCH_FAIL_ATTEMPT="0", SL_COMM_TIMEOUT_MS="4000")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--rows", "200000", "--steps", "3", "--warmup", "1", "--no-sweep"], cwd=ROOT,
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    a = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    att = a["config"]["launcher"]["attempts"]
    assert [x["ok"] for x in att] == [False, True] and a["config"]["exchange_verified"] is True


def test_partitioned_solve_script_two_ranks_equals_one(gpu):
    """tools/solve_partitioned.py (BASELINE config 5 as a full solve): two ranks sharing the GPU run the same iterations to the
    same residual and solution sums as one rank owning all rows"""
    import os
    import socket

    def run(ranks, rows, bandwidth=700, bounds=None, exchange="torch"):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
               "--master-port", str(port), "tools/solve_partitioned.py", "--rows", str(rows), "--bandwidth", str(bandwidth), "--tolerance", "1e-9"]
        if bounds:
            cmd += ["--bounds", bounds]
        cmd += ["--exchange", exchange]
        r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=dict(os.environ, SL_BENCH_BACKEND="gloo"))
        assert r.returncode == 0, r.stderr[-3000:]
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])

    one, two = run(1, 200_000), run(2, 100_000)
    assert one["converged"] and two["converged"] and two["n_gpus"] == 2
    assert (one["iterations"], one["terms"]) == (two["iterations"], two["terms"])
    for key in ("residual_norm", "sum_x", "sum_x2", "last_term_norm"):
        assert abs(one[key] - two[key]) <= 1e-11 * max(1.0, abs(one[key])), key
    # the same solve through the library's own communicator (C ABI), two ranks sharing the GPU, equal and unequal ranges
    for kw in (dict(rows=100_000), dict(rows=0, bounds="0,61234,200000")):
        abi = run(2, exchange="abi", **kw)
        assert abi["converged"] and (one["iterations"], one["terms"]) == (abi["iterations"], abi["terms"]) and "abi" in abi["config"]
        for key in ("residual_norm", "sum_x", "sum_x2", "last_term_norm"):
            assert abs(one[key] - abi[key]) <= 1e-11 * max(1.0, abs(one[key])), key
    # unequal row ranges (RowPartition with explicit bounds, as nnz_balanced_bounds returns): halo and gather exchanges
    for bw in (700, 0):
        ref = one if bw == 700 else run(1, 200_000, bw)
        skew = run(2, 0, bw, bounds="0,61234,200000")
        assert skew["converged"] and (ref["iterations"], ref["terms"]) == (skew["iterations"], skew["terms"])
        for key in ("residual_norm", "sum_x", "sum_x2", "last_term_norm"):
            assert abs(ref[key] - skew[key]) <= 1e-11 * max(1.0, abs(ref[key])), (bw, key)


def test_bench_initialises_rccl_on_this_box(gpu):
    """world size 1 through the real backend ("nccl" = RCCL): process-group init with device_id, barrier and all-reduce run
    through RCCL on the GPU box (the halo transfer itself needs a second GPU and is first exercised by the multi-GPU run)"""
    import os
    r = subprocess.run([sys.executable, "bench.py", "--rows", "300000", "--k", "16", "--bandwidth", "512", "--steps", "5", "--warmup", "1",
                        "--no-cpu-baseline", "--no-sweep", "--force-split"], cwd=ROOT, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, SL_BENCH_FORCE_DIST="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1"))
    assert r.returncode == 0, r.stderr[-3000:]
    a = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert a["n_gpus"] == 1 and a["config"]["exchange"] == "halo+overlap" and a["value"] > 0
