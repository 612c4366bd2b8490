#!/bin/bash
# First contact with a box that has N >= 2 GPUs: the first execution EVER of the library's RCCL transport with more
# than one rank, and of the ipc transport across two physical devices.  Every stage has its own timeout and its own log under
# gpurun_out/first_contact/; a stage that fails does not stop the later ones; the last line is a one-line verdict per stage.
#   tools/first_contact.sh [N = number of visible GPUs] [rows per rank for the bench stage = 10000000]
# Stages: (i) tests/c/dist_smoke.c (C99, one process per rank, every solution entry compared BIT FOR BIT with the one-GPU solve through the
#             same ABI) at world N under SL_COMM_TRANSPORT=ipc, then rccl with SL_COMM_HALO=sendrecv (grouped ncclSend / ncclRecv) and
#             =allreduce (one ncclAllReduce over the compact halo buffer: BASELINE north_star's wording), each on a neighbour-halo system,
#             an all-over-the-matrix system (ncclAllGather) and an uneven three-way reach;
#         (ii) python bench.py --gpus N (the line the driver's SCALE run asks for), once per transport;
#         (iii) the same line with both exchanges measured in one job (exchange_variants), self-launched and under torch.distributed.run.
# One job at a time, N processes, nothing side by side.
#
# WHAT TO LOOK AT FIRST.  Until this script runs, the library's RCCL calls have met more than one rank only against the test suite's
# stand-in (tests/simt/fake_rccl.cpp), which implements what the code ASSUMES of RCCL.  Three assumptions a real librccl may not share —
# if a `rccl` stage fails while its `ipc` twin passes, read its log with these in mind (SL_LOG=1 names the call that returned the error):
#   1. IN-PLACE ncclAllGather: sl_comm.hip passes sendbuff = recvbuff + rank * count (the rank's own rows inside the gathered vector).
#      NCCL documents exactly this aliasing as the in-place form; the stand-in accepts any overlap.  A real "invalid argument" here means
#      the offset arithmetic (rows per rank must be equal: `equal_ranges`) is off by something the stand-in does not check.
#   2. ZERO-LENGTH members of a group: the halo exchange is ONE ncclGroupStart / End around a Send and a Recv per peer, and a peer whose
#      strip is empty (uneven reach, stage *_uneven) is SKIPPED on both sides by the same rule (give / need lists built from the same
#      all-gathered reaches).  The stand-in matches sends to receives by (peer, order); real RCCL hangs — it does not fail — when one
#      side posts a member the other skipped: a stage that ends in SL_COMM_TIMEOUT_MS with "ncclGroupEnd" last in the log is this.
#   3. STREAM ORDER against the side stream of the boundary-first step: the collective is enqueued on the exchange stream after an event
#      recorded behind the edge blocks on the launch stream; the stand-in runs every call synchronously, so ordering bugs cannot show
#      there.  Symptom on hardware: `differ` in dist_smoke (a stale halo), only with SL_DIST_OVERLAP unset, gone with SL_DIST_OVERLAP=0.
#   Also: ncclAllReduce (SL_COMM_HALO=allreduce) sums a buffer that holds -0.0 wherever a rank exports nothing, so that x + (-0.0) keeps
#   every bit of x; a reduction that flushes -0.0 or reorders into a tree changes nothing here (each slot has exactly one non-zero term).
set -u
cd "$(dirname "$0")/.."
NDEV=$(python - <<'PY'
import ctypes, sys
sys.path.insert(0, ".")
from sublinear_time_solver_amd import _lib
n = ctypes.c_int(0); _lib.load().sl_device_count(ctypes.byref(n)); print(n.value)
PY
)
N=${1:-$NDEV}; ROWS=${2:-10000000}
O=gpurun_out/first_contact; mkdir -p $O; : > $O/verdict.txt
echo "first contact: $NDEV devices visible, world $N" | tee -a $O/verdict.txt
if [ "$NDEV" -lt 2 ] || [ "$N" -lt 2 ]; then echo "needs 2+ GPUs: nothing run" | tee -a $O/verdict.txt; exit 3; fi
PKG=$PWD/sublinear_time_solver_amd
gcc -std=c99 -pedantic -Wall -Iinclude tests/c/dist_smoke.c -o /tmp/dist_smoke_fc -L$PKG -lsublinear_hip -lm -Wl,-rpath,$PKG || { echo "dist_smoke did not build" | tee -a $O/verdict.txt; exit 1; }
stage() {      # stage <name> <timeout> <command...>: log, exit status and the decisive lines into verdict.txt
  local name=$1 tmo=$2; shift 2
  timeout $tmo "$@" > $O/$name.log 2>&1; local rc=$?
  printf '%-44s rc %3d  %s\n' "$name" $rc "$(grep -E 'dist_smoke ok|differ|"metric"' $O/$name.log | head -n 1 | cut -c1-220)" | tee -a $O/verdict.txt
}
for T in "ipc sendrecv" "rccl sendrecv" "rccl allreduce"; do
  set -- $T; TR=$1; HALO=$2
  export SL_LOG=1 SL_COMM_TRANSPORT=$TR SL_COMM_TIMEOUT_MS=60000
  if [ "$HALO" = allreduce ]; then export SL_COMM_HALO=allreduce; else unset SL_COMM_HALO; fi
  stage smoke_${TR}_${HALO}_halo      300 /tmp/dist_smoke_fc $N 400000 300
  stage smoke_${TR}_${HALO}_allgather 300 /tmp/dist_smoke_fc $N 400000 1000000000
  stage smoke_${TR}_${HALO}_uneven    300 /tmp/dist_smoke_fc $N 3000000 4096 uneven
done
unset SL_COMM_TRANSPORT SL_COMM_HALO SL_LOG
for TR in ipc rccl; do
  SL_BENCH_TRANSPORTS=$TR stage bench_${TR} 1500 python bench.py --gpus $N --rows $ROWS --steps 30 --warmup 5
done
# (iii) the line as the driver's scaling run gets it: both exchanges in one job (exchange_variants: ipc, and rccl + ONE all-reduce over the halo), once
#       self-launched and once under torch.distributed.run
stage bench_both_selflaunched 1700 python bench.py --gpus $N --rows $ROWS --steps 30 --warmup 5
stage bench_both_torchrun 1700 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus $N --rows $ROWS --steps 30 --warmup 5
echo "logs: $O/*.log"; cat $O/verdict.txt
grep -c "rc   0" $O/verdict.txt
