#!/bin/bash
# A/B of band-kernel block sizes on the GPU box (gpurun -- 'bash tools/ab.sh'): 4-, 8- and 16-wave blocks at several window
# widths.  Repetitions are interleaved because sustained clocks move a configuration by +-5 % between runs; compare inside
# one repetition.  (profiles/r01_ab_wave_blocks.txt is the first such run, taken with two builds of the library.)
mkdir -p gpurun_out
P='import sys,json; d=json.loads(sys.stdin.readline()); print(d["config"]["half_bandwidth"], round(d["ms_per_step"],4), round(d["roofline"]["launch_ms"],4), round(d["roofline"]["frac"],4))'
run() { python bench.py --steps 60 --warmup 5 --bandwidth $1 --no-cpu-baseline --no-sweep 2>&1 | grep metric | python -c "$P"; }
{
for rep in 1 2 3; do
for w in 512 1024 2048 4096; do
for nw in 4 8 16; do echo -n "w$w nw=$nw "; SL_BAND_NW=$nw run $w; done
done
for w in 6000 8192; do echo -n "w$w nw=16 "; run $w; echo -n "w$w general "; SL_BAND_NW16=0 run $w; done
done
} > gpurun_out/ab_blocks.txt 2>&1
cat gpurun_out/ab_blocks.txt
