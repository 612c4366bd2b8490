# tools/ab_small_paced.sh: where the paced panel kernel starts to pay on uniform columns (general kernel against paced, auto lead and lead 2)
cd /root/repo
run() { python bench.py --n $1 --k $2 --bandwidth 0 --no-sweep --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('   n', d['config']['n_per_gpu'], 'k', d['config']['nnz_per_row'], d['roofline']['kernel'][:24], 'ms', round(d['roofline']['launch_ms'],4), 'frac', round(d['roofline']['frac'],3))"; }
for nk in "300000 8" "300000 16" "500000 8" "500000 16" "700000 8" "700000 16" "1500000 8" "1500000 16" "3000000 8" "5000000 16"; do set -- $nk
  echo general; SL_COLUMN_PANELS=0 run $1 $2
  echo "paced auto"; SL_COLUMN_PANELS=1 SL_PW_FORCE=1 run $1 $2
  echo "paced slack 2"; SL_COLUMN_PANELS=1 SL_PW_FORCE=1 SL_PW_SLACK=2 run $1 $2
  echo "default"; run $1 $2
done
