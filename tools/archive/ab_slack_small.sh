cd /root/repo
run() { python bench.py --n $1 --k $2 --bandwidth 0 --no-sweep --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('   n', d['config']['n_per_gpu'], 'k', d['config']['nnz_per_row'], 'ms', round(d['roofline']['launch_ms'],4))"; }
for nk in "1000000 8" "1000000 16" "3000000 8" "5000000 16" "10000000 8" "20000000 16"; do set -- $nk
  for sl in 1 2 3; do echo "slack $sl"; SL_COLUMN_PANELS=1 SL_PW_FORCE=1 SL_PW_SLACK=$sl run $1 $2; done
done
