# tools/ab_c2_paced.sh: n = 10^6 (BASELINE config 1's size), uniform columns: general kernel against the paced panel kernel at several leads
cd /root/repo
run() { python bench.py --n $1 --k $2 --bandwidth 0 --no-sweep --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('   n', d['config']['n_per_gpu'], 'k', d['config']['nnz_per_row'], d['roofline']['kernel'][:24], 'ms', round(d['roofline']['launch_ms'],4), 'frac', round(d['roofline']['frac'],3))"; }
for nk in "1000000 8" "1000000 16" "2000000 8"; do set -- $nk
  echo general; SL_COLUMN_PANELS=0 run $1 $2
  for sl in 0 2 8 32 1048576; do echo "paced slack $sl"; SL_COLUMN_PANELS=1 SL_PW_FORCE=1 SL_PW_SLACK=$sl run $1 $2; done
done
