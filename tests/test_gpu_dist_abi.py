"""Multi-GPU behind the C ABI: tests/c/dist_smoke.c — strict C99, no Python in the solve — forks one process per
rank (the ranks share the box's GPU), partitions a seeded system by rows, and solves it through sl_comm_* /
sl_neumann_state_create_partitioned / _update_rhs / _reset / _run / _solution; the parent compares with the one-GPU solve through
the same ABI: same iteration count, same convergence flag, solution bit for bit."""
import os
import subprocess
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def dist_exe(tmp_path_factory):
    exe = tmp_path_factory.mktemp("dist") / "dist_smoke"
    pkg, name = ROOT / "sublinear_time_solver_amd", "sublinear_hip"
    simt = Path(os.environ.get("SUBLINEAR_HIP_LIB", ""))
    if "simt" in simt.name:                 # tests/test_simt_emulated.py: the same program against the emulator library (no GPU in the container)
        pkg, name = simt.parent, simt.name[3:].split(".")[0]
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", f"-I{ROOT / 'include'}", str(ROOT / "tests" / "c" / "dist_smoke.c"),
                        "-o", str(exe), f"-L{pkg}", f"-l{name}", "-lm", f"-Wl,-rpath,{pkg}"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


@pytest.mark.parametrize("world,n,w,uneven", [(2, 20000, 300, False),       # neighbour halo only
                                              (3, 50000, 700, True),        # three ranks, unequal row ranges
                                              (2, 30000, 10**9, False),     # columns all over the matrix: every rank pulls everything
                                              (4, 40000, 15000, True),      # reach beyond the next neighbour
                                              (3, 1500000, 10**9, True),    # 8 MB per exchange: the pulls run on one stream per peer
                                              (1, 5000, 50, False)])        # a communicator of one
def test_partitioned_solve_through_the_c_abi(gpu, dist_exe, world, n, w, uneven):
    env = dict(os.environ, SL_COMM_TIMEOUT_MS="30000")
    r = subprocess.run([str(dist_exe), str(world), str(n), str(w)] + (["uneven"] if uneven else []), capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "dist_smoke ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.parametrize("world,n,w,overlap,expect", [(2, 200000, 300, "1", True),      # edge blocks first, exchange beside the interior
                                                      (2, 200000, 300, "0", False),     # the same system, exchange after the whole step
                                                      (3, 600000, 5000, "1", True),     # band kernel with a wide window, unequal ranges
                                                      (4, 40000, 15000, "1", False),    # reach beyond the neighbour: no interior to hide behind
                                                      (1, 300000, 2000, "2", True)])    # forced on at world size 1 (no peers, same stream topology)
def test_boundary_first_step_gives_the_same_bits(gpu, dist_exe, world, n, w, overlap, expect):
    """the partitioned step with its edge blocks, halo ticket and pulls on the side stream beside the interior blocks: same iteration
    count, same solution bit for bit as the one-GPU solve (checked inside dist_smoke), and the log says which form ran"""
    env = dict(os.environ, SL_COMM_TIMEOUT_MS="30000", SL_DIST_OVERLAP=overlap, SL_LOG="1")
    r = subprocess.run([str(dist_exe), str(world), str(n), str(w)] + (["uneven"] if world == 3 else []), capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "dist_smoke ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    assert ("runs its edge blocks first" in r.stderr) == expect, r.stderr[-3000:]


@pytest.mark.parametrize("world,n,w,forced,expect", [(2, 1_200_000, 30_000, True, True),        # pretended 8-CU device with 4 L2 groups: 4 rounds, the first one = the edge
                                                     (4, 3_000_000, 60_000, True, True),
                                                     (3, 2_000_000, 25_000, True, False),       # unequal ranges: one rank has a single round, so nobody splits
                                                     (2, 20_000_000, 2_500_000, False, True)])  # SURVEY 8(e)'s C5 structure at its own size per rank: w = n_local / 4
def test_paced_layout_runs_its_edge_rounds_first(gpu, dist_exe, world, n, w, forced, expect):
    """locality-bounded columns far beyond the L2 (the paced layout with XCD-local spans): a rank's rows within w of its range ends sit in
    the spans of the FIRST rounds; the partitioned step launches those rounds, hands the halo ticket out and pulls beside the remaining
    rounds.  Same iteration count and solution bits as the one-GPU solve (checked inside dist_smoke), with the split and without it."""
    env = dict(os.environ, SL_COMM_TIMEOUT_MS="120000", SL_LOG="1")
    if forced:
        env.update(SL_COLUMN_PANELS="1", SL_PW_FORCE="1", SL_PW_CUS="8", SL_PW_XCD="4")
    for overlap in ("1", "0"):
        r = subprocess.run([str(dist_exe), str(world), str(n), str(w)] + (["uneven"] if world == 3 else []), capture_output=True, text=True, timeout=600,
                           env=dict(env, SL_DIST_OVERLAP=overlap))
        assert r.returncode == 0 and "dist_smoke ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
        assert "paced column panels" in r.stderr
        assert ("runs its edge rounds first" in r.stderr) == (expect and overlap == "1"), r.stderr[-3000:]


@pytest.mark.parametrize("halo", ["", "allreduce"])
def test_rccl_transport_of_the_library_at_world_one(gpu, dist_exe, halo):
    """SL_COMM_TRANSPORT=rccl: librccl resolved at run time, ncclCommInitRank with the id passed through the shared block, the
    partial sums through ncclAllGather + the rank-order sum kernel — everything of the transport that one GPU can run (RCCL refuses
    two ranks on one device; the first multi-GPU box runs tools/ab_transport.sh).  Same bits as the one-GPU solve."""
    env = dict(os.environ, SL_COMM_TIMEOUT_MS="30000", SL_COMM_TRANSPORT="rccl", SL_LOG="1")
    if halo:
        env["SL_COMM_HALO"] = halo
    r = subprocess.run([str(dist_exe), "1", "60000", "400"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "dist_smoke ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    assert "transport rccl" in r.stderr, r.stderr[-2000:]


def test_two_ranks_on_one_device_are_refused_by_the_rccl_transport_collectively(gpu, dist_exe):
    """RCCL needs one rank per GPU: on the one-GPU box ncclCommInitRank fails — on every rank, with an error, within the time limit
    (the agreement after the init), never with a hang or a half-built communicator"""
    env = dict(os.environ, SL_COMM_TIMEOUT_MS="20000", SL_COMM_TRANSPORT="rccl")
    r = subprocess.run([str(dist_exe), "2", "20000", "300"], capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode != 0 and "dist_smoke ok" not in r.stdout, r.stdout[-2000:]


def test_a_rank_that_never_arrives_becomes_an_error_not_a_hang(gpu):
    """world = 2 with only rank 0 present: the rendezvous wait is bounded (SL_COMM_TIMEOUT_MS) and comes back as DeviceError"""
    import sys
    code = ("import sys, time; sys.path.insert(0, %r); import sublinear_time_solver_amd as S\n"
            "t0 = time.time()\n"
            "try:\n    S.Communicator(0, 2, 'lonely_rank_test')\n    print('NO ERROR')\n"
            "except S.SolverError as e:\n    print('kind', e.kind, 'after', round(time.time() - t0, 1))\n") % str(ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=dict(os.environ, SL_COMM_TIMEOUT_MS="1500"))
    assert r.returncode == 0 and "kind DeviceError" in r.stdout, r.stdout + r.stderr[-1500:]
    assert float(r.stdout.split("after")[1]) < 30.0


def test_a_failure_on_one_rank_is_the_verdict_of_all_and_the_communicator_stays_in_step(gpu, tmp_path):
    """rank 1's rows are not diagonally dominant: NeumannState::new over the partition fails on BOTH ranks with that status (the
    agreement after the local stage), nobody waits for anybody, and the SAME communicator then builds and solves a valid system —
    its barrier / exchange counters are still in step on every rank (collective construction must not fall out of step)"""
    import sys
    script = tmp_path / "rank.py"
    script.write_text('''
import sys, time, numpy as np
sys.path.insert(0, %r)
import sublinear_time_solver_amd as S
from sublinear_time_solver_amd import generators as G
rank, name = int(sys.argv[1]), sys.argv[2]
n, k = 40000, 8
lo, hi = rank * n // 2, (rank + 1) * n // 2
rp, ci, va, b = G.sdd_rows(n, k, 3, 200, lo, hi)
comm = S.Communicator(rank, 2, name)
bad = va.copy()
if rank == 1:
    bad[ci[:] == (lo + np.repeat(np.arange(hi - lo), k))] = 1e-3          # diagonal far below the row sums
mb = S.SparseMatrix.from_csr(rp, ci, bad, hi - lo, n, row_offset=lo)
t0 = time.time()
try:
    S.NeumannSolver().initialize_partitioned(comm, mb, b, S.SolverOptions(tolerance=1e-10))
    print("NO ERROR")
except S.SolverError as e:
    print("first", e.kind, round(time.time() - t0, 2))
m = S.SparseMatrix.from_csr(rp, ci, va, hi - lo, n, row_offset=lo)
st = S.NeumannSolver().initialize_partitioned(comm, m, b, S.SolverOptions(tolerance=1e-10))
r = st.run()
print("second", r.converged, r.iterations, float(np.abs(r.solution).sum()))
st.close(); comm.close()
''' % str(ROOT))
    name = f"agree_{os.getpid()}"
    env = dict(os.environ, SL_COMM_TIMEOUT_MS="20000")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), name], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in (0, 1)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so + se[-2000:]
        assert "first MatrixNotDiagonallyDominant" in so and "NO ERROR" not in so, so + se[-1500:]
        assert float(so.split("first MatrixNotDiagonallyDominant")[1].split()[0]) < 10.0              # nobody ran into a time limit
        assert "second True" in so, so + se[-1500:]
    it = {so.split("second True")[1].split()[0] for so, _ in outs}
    assert len(it) == 1                                                                               # the same iteration count on both ranks
