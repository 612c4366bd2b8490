#!/bin/bash
# First thing to run on a box with 2+ GPUs: the partitioned solve (tests/c/dist_smoke.c: C99, one process per rank, solution compared
# bit for bit with the one-GPU solve, exchange verified against the owners' copies) and bench.py under each transport of the library.
#   tools/ab_transport.sh [ranks=2] [rows per rank for bench=10000000]
# Writes gpurun_out/ab_transport.txt.
set -u
cd "$(dirname "$0")/.."
N=${1:-2}; ROWS=${2:-10000000}
OUT=gpurun_out/ab_transport.txt; mkdir -p gpurun_out; : > $OUT
PKG=$PWD/sublinear_time_solver_amd
gcc -std=c99 -pedantic -Wall -Iinclude tests/c/dist_smoke.c -o /tmp/dist_smoke -L$PKG -lsublinear_hip -lm -Wl,-rpath,$PKG || exit 1
for T in ipc rccl "rccl allreduce"; do
  set -- $T; TR=$1; HALO=${2:-}
  # (the last case = config 5's locality-bounded form at its own size per rank: paced layout, XCD-local spans, edge rounds first)
  for CASE in "$N 400000 300" "$N 400000 1000000000" "$N 3000000 4096 uneven" "$N $((N * 10000000)) 2500000"; do
    echo "== dist_smoke $CASE  transport=$TR halo=${HALO:-sendrecv}" | tee -a $OUT
    SL_LOG=1 SL_COMM_TRANSPORT=$TR SL_COMM_HALO=$HALO SL_COMM_TIMEOUT_MS=60000 timeout 600 /tmp/dist_smoke $CASE 2>&1 | grep -E "dist_smoke ok|differ|failed|exited|transport|edge blocks|edge rounds" | sort | uniq -c | tee -a $OUT
  done
  echo "== bench --gpus $N --rows $ROWS transport=$TR halo=${HALO:-sendrecv}" | tee -a $OUT
  SL_BENCH_TRANSPORTS=$TR SL_COMM_HALO=$HALO timeout 1800 python bench.py --gpus $N --rows $ROWS --steps 30 --warmup 5 2>>$OUT | tee -a $OUT
done
echo "== bench --gpus $N transport=ipc, exchange AFTER the whole step (SL_DIST_OVERLAP=0): what the edge-blocks / edge-rounds-first step hides" | tee -a $OUT
SL_BENCH_TRANSPORTS=ipc SL_DIST_OVERLAP=0 timeout 1800 python bench.py --gpus $N --rows $ROWS --steps 30 --warmup 5 2>>$OUT | tee -a $OUT
echo "== bench --gpus $N over torch.distributed (exchange above the ABI)" | tee -a $OUT
SL_BENCH_TRANSPORTS=torch timeout 1800 python bench.py --gpus $N --rows $ROWS --steps 30 --warmup 5 2>>$OUT | tee -a $OUT
