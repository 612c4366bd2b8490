cd /root/repo
run() { python bench.py --bandwidth 4096 --no-sweep --no-cpu-baseline --steps 50 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print(d['config']['exchange'][:20], 'ms/step', round(d['ms_per_step'],4), 'dev', round(d['roofline']['launch_ms'],4))"; }
for i in 1 2; do echo plain; run; echo abi-world1; SL_BENCH_FORCE_ABI=1 run; done
