#!/bin/bash
# A/B of two builds of the library on the headline configuration (interleaved repetitions; see sweep.sh)
mkdir -p gpurun_out
P='import sys,json; d=json.loads(sys.stdin.readline()); print(d["config"]["half_bandwidth"], round(d["ms_per_step"],4), round(d["roofline"]["launch_ms"],4), round(d["roofline"]["frac"],4))'
run() { python bench.py --steps 60 --warmup 5 --bandwidth $1 --no-cpu-baseline --no-sweep 2>&1 | grep metric | python -c "$P"; }
B=$PWD/sublinear_time_solver_amd/libsublinear_hip_b512.so
{
for rep in 1 2 3; do
for w in 4096 512; do
echo -n "w$w base "; run $w
echo -n "w$w b512 "; SUBLINEAR_HIP_LIB=$B run $w
for spw in 2 3 4; do echo -n "w$w b512 spw=$spw "; SUBLINEAR_HIP_LIB=$B SL_BAND_SPW=$spw run $w; done
done
done
} > gpurun_out/ab_b512.txt 2>&1
cat gpurun_out/ab_b512.txt
