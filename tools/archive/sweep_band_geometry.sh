# tools/sweep_band_geometry.sh: waves per block x slices per wave of the band kernel on the uniform-width headline shape (k = 16)
cd /root/repo
run() { python bench.py --bandwidth $1 --no-sweep --no-cpu-baseline --steps 60 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('   w', d['config']['half_bandwidth'], 'ms', round(d['roofline']['launch_ms'],4), 'frac', round(d['roofline']['frac'],3))"; }
for w in 512 4096; do
  echo "default"; run $w
  for nw in 4 8; do for spw in 2 3 4 6 8; do echo "NW=$nw SPW=$spw"; SL_BAND_NW=$nw SL_BAND_SPW=$spw run $w; done; done
  echo "NW=16 SPW=4"; SL_BAND_NW=16 SL_BAND_SPW=4 run $w
  echo "default"; run $w
done
