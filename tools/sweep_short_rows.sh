cd /root/repo
run() { python bench.py --k $1 --bandwidth $2 --no-sweep --no-cpu-baseline --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('   k', d['config']['nnz_per_row'], 'w', d['config']['half_bandwidth'], 'ms', round(d['roofline']['launch_ms'],4), 'frac', round(d['roofline']['frac'],3))"; }
for k in 5 9; do for w in 512 4096; do
  echo "default"; run $k $w
  for nw in 4 8; do for spw in 2 4 6 8; do echo "NW=$nw SPW=$spw"; SL_BAND_NW=$nw SL_BAND_SPW=$spw run $k $w; done; done
done; done
