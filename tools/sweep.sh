mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -5
P='import sys,json; d=json.loads(sys.stdin.readline()); print(d["config"]["half_bandwidth"], round(d["ms_per_step"],4), round(d["roofline"]["launch_ms"],4), round(d["roofline"]["frac"],4))'
for c16 in 1 0; do for pipe in 0 1; do for spw in 2 4; do for w in 512 4096; do echo -n "c16=$c16 pipe=$pipe spw=$spw "; SL_BAND_C16=$c16 SL_BAND_PIPE=$pipe SL_BAND_SPW=$spw python bench.py --steps 30 --warmup 3 --bandwidth $w --no-cpu-baseline --no-sweep 2>&1 | grep metric | python -c "$P"; done; done; done; done > gpurun_out/sweep4.txt 2>&1
cat gpurun_out/sweep4.txt
