mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/pytest2.txt
cat gpurun_out/pytest2.txt
P='import sys,json; d=json.loads(sys.stdin.readline()); print(d["config"]["half_bandwidth"], round(d["ms_per_step"],4), round(d["roofline"]["launch_ms"],4), round(d["roofline"]["frac"],4))'
for spw in 2 4 8; do for w in 64 512 2048 4096; do echo -n "spw=$spw "; SL_BAND_SPW=$spw python bench.py --steps 30 --warmup 3 --bandwidth $w --no-cpu-baseline 2>/dev/null | python -c "$P"; done; done > gpurun_out/sweep2.txt 2>&1
for w in 64 4096; do echo -n "general "; SL_BAND_DISABLE=1 python bench.py --steps 30 --warmup 3 --bandwidth $w --no-cpu-baseline 2>/dev/null | python -c "$P"; done >> gpurun_out/sweep2.txt 2>&1
cat gpurun_out/sweep2.txt
