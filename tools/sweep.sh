#!/bin/bash
# A/B sweep of the band-kernel geometry on the GPU box (gpurun -- 'bash tools/sweep.sh').  Repetitions are
# interleaved because DVFS moves a single configuration by +-5 % between runs; compare within a repetition.
mkdir -p gpurun_out
P='import sys,json; d=json.loads(sys.stdin.readline()); print(d["config"]["half_bandwidth"], round(d["ms_per_step"],4), round(d["roofline"]["launch_ms"],4), round(d["roofline"]["frac"],4))'
run() { python bench.py --steps 60 --warmup 5 --bandwidth $1 --no-cpu-baseline --no-sweep 2>&1 | grep metric | python -c "$P"; }
{
for rep in 1 2 3; do
for w in 4096 512; do
for spw in 2 4 6; do echo -n "w$w spw=$spw "; SL_BAND_SPW=$spw run $w; done
echo -n "w$w pipe=0 "; SL_BAND_PIPE=0 run $w
echo -n "w$w c16=0 "; SL_BAND_C16=0 run $w
echo -n "w$w general "; SL_BAND_DISABLE=1 run $w
done
done
} > gpurun_out/sweep_band.txt 2>&1
cat gpurun_out/sweep_band.txt
