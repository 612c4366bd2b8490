cd /root/repo
run() { python bench.py --bandwidth $1 --k $2 --no-sweep --no-cpu-baseline --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('   w', d['config']['half_bandwidth'], 'k', d['config']['nnz_per_row'], d['roofline']['kernel'][:16], 'ms', round(d['roofline']['launch_ms'],4), 'frac', round(d['roofline']['frac'],3))"; }
for wk in "100000 16" "300000 16" "1000000 16" "32768 8" "20000 16" "60000 16"; do set -- $wk
  echo default; run $1 $2
  for b in 9 10; do for sl in 1 2 3; do echo "2^$b slack $sl"; SL_PW_BAND=$b SL_PW_SLACK=$sl run $1 $2; done; done
done
