#!/bin/bash
# What the batched tile epilogue is worth on the headline step: the library rebuilt with SL_PW_EPI_ROWS = 1 (one row per trip,
# as until round 6 but without the per-trip vmcnt(0)), 4 (the default), 8, each timed by bench.py's own launch timer, twice, interleaved.
# (8 is a measurement build only: the headline instantiation fits 128 VGPRs without a spill, the push instantiations do not.)
# Runs on the GPU box (hipcc is there: ~1 min per build); leaves the default build behind.  usage: bash tools/ab_epi_rows.sh
cd "$(dirname "$0")/.."
run() { python bench.py --steps 30 --no-sweep --no-cpu-baseline --no-parity-gate 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('SL_PW_EPI_ROWS=$1', 'launch_ms', round(d['roofline']['launch_ms'],4), 'ms_per_step', round(d['ms_per_step'],4))"; }
for rep in 1 2; do
  for r in 1 4 8; do
    touch sublinear_time_solver_amd/csrc/sl_kernels.hip
    make -s -C sublinear_time_solver_amd/csrc EXTRA=-DSL_PW_EPI_ROWS=$r > /dev/null 2>&1 || { echo "build with SL_PW_EPI_ROWS=$r failed"; continue; }
    run $r
  done
done
touch sublinear_time_solver_amd/csrc/sl_kernels.hip; make -s -C sublinear_time_solver_amd/csrc > /dev/null 2>&1
