# tools/ab_band_auto.sh: wide bands, the general kernel (SL_PW_BAND=0) against the auto-selected layout
cd /root/repo
run() { python bench.py --n $1 --bandwidth $2 --k $3 --no-sweep --no-cpu-baseline --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('   n', d['config']['n_per_gpu'], 'w', d['config']['half_bandwidth'], 'k', d['config']['nnz_per_row'], d['roofline']['kernel'][:16], 'ms', round(d['roofline']['launch_ms'],4), 'frac', round(d['roofline']['frac'],3))"; }
for c in "10000000 9000 16" "10000000 12000 16" "10000000 20000 16" "10000000 32768 16" "10000000 60000 16" "10000000 100000 16" "10000000 200000 16" "10000000 300000 16" "10000000 32768 8" "10000000 32768 5" "3000000 32768 16" "1500000 20000 16" "1048576 32768 16"; do set -- $c
  echo general; SL_PW_BAND=0 run $1 $2 $3
  echo auto; run $1 $2 $3
done
