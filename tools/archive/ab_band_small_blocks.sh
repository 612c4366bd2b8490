# tools/ab_band_small_blocks.sh: band kernel, default geometry against 8 waves x 2 slices, interleaved
cd /root/repo
run() { python bench.py --bandwidth $1 --k $2 --no-sweep --no-cpu-baseline --steps 60 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('   w', d['config']['half_bandwidth'], 'k', d['config']['nnz_per_row'], 'ms', round(d['roofline']['launch_ms'],4), 'frac', round(d['roofline']['frac'],3))"; }
for i in 1 2 3; do for wk in "512 16" "128 16" "4096 16" "1024 16" "512 8" "2048 16"; do set -- $wk
  echo "default"; run $1 $2
  echo "8x2"; SL_BAND_NW=8 SL_BAND_SPW=2 run $1 $2
done; done
