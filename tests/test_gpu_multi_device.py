"""Runs only on a box with two or more GPUs (skipped on the one-GPU boxes this repository has met so far): the first execution of the
library's transports ACROSS devices — SL_COMM_TRANSPORT=ipc over xGMI-mapped peer memory, and SL_COMM_TRANSPORT=rccl with more than one
rank (grouped ncclSend / ncclRecv halo strips; SL_COMM_HALO=allreduce: one ncclAllReduce over the compact halo buffer, BASELINE
north_star's form; ncclAllGather of the term for columns all over the matrix).  tests/c/dist_smoke.c compares every entry of the
partitioned solution BIT FOR BIT with the one-GPU solve through the same ABI.  tools/first_contact.sh is the same thing as a script with
per-stage logs (+ bench.py --gpus N).  Precedent for row chunks behind one call: src/simd_ops.rs:201-239; SURVEY 8(e)."""
import ctypes
import os
import subprocess
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _devices():
    from sublinear_time_solver_amd import _lib
    n = ctypes.c_int(0)
    _lib.load().sl_device_count(ctypes.byref(n))
    return n.value


@pytest.fixture(scope="module")
def dist_exe(tmp_path_factory):
    exe = tmp_path_factory.mktemp("dist_md") / "dist_smoke"
    pkg = ROOT / "sublinear_time_solver_amd"
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", f"-I{ROOT / 'include'}", str(ROOT / "tests" / "c" / "dist_smoke.c"),
                        "-o", str(exe), f"-L{pkg}", "-lsublinear_hip", "-lm", f"-Wl,-rpath,{pkg}"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


@pytest.mark.parametrize("transport,halo", [("ipc", ""), ("rccl", ""), ("rccl", "allreduce")])
@pytest.mark.parametrize("n,w,uneven", [(400000, 300, False),          # neighbour halo strips
                                        (400000, 10**9, False),       # every rank needs every row: all-gather
                                        (3000000, 4096, True)])       # unequal row ranges
def test_one_rank_per_gpu_gives_the_one_gpu_bits(gpu, dist_exe, transport, halo, n, w, uneven):
    ndev = _devices()
    if ndev < 2:
        pytest.skip(f"{ndev} GPU visible: the multi-device transports need two or more (tools/first_contact.sh on the first such box)")
    world = min(ndev, 8)
    env = dict(os.environ, SL_COMM_TIMEOUT_MS="60000", SL_COMM_TRANSPORT=transport, SL_LOG="1")
    env.pop("SL_COMM_HALO", None)
    if halo:
        env["SL_COMM_HALO"] = halo
    r = subprocess.run([str(dist_exe), str(world), str(n), str(w)] + (["uneven"] if uneven else []), capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "dist_smoke ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    assert f"transport {transport}" in r.stderr, r.stderr[-1500:]


def test_bench_line_at_every_visible_gpu(gpu):
    ndev = _devices()
    if ndev < 2:
        pytest.skip(f"{ndev} GPU visible")
    import json
    import sys
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", str(min(ndev, 8)), "--rows", "2000000", "--steps", "10", "--warmup", "2", "--no-sweep",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')][-1])
    assert line["n_gpus"] == min(ndev, 8) and line["scaling"] == "weak" and line["value"] > 0
    assert line["parity_gate"]["bitwise_equal"] is True
