cd /root/repo
run() { python bench.py --n $1 --k 16 --bandwidth $2 --no-sweep --no-cpu-baseline --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('n', d['config']['n_per_gpu'], 'w', d['config']['half_bandwidth'], d['roofline']['kernel'][:34], 'ms', round(d['roofline']['launch_ms'],4), 'frac', round(d['roofline']['frac'],3))"; }
for w in 312500 1000000 4000000; do echo "default w=$w"; run 10000000 $w; echo "no panels"; SL_COLUMN_PANELS=0 run 10000000 $w; done
run 1000000 0; run 2000000 0; run 3000000 0
