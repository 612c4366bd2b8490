# tools/ab_band_paced.sh: wide bands — the general kernel against the paced layout with block-local rows and narrow panels (SL_PW_BAND = log2 width)
cd /root/repo
run() { python bench.py --bandwidth $1 --no-sweep --no-cpu-baseline --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('   w', d['config']['half_bandwidth'], d['roofline']['kernel'][:26], 'ms', round(d['roofline']['launch_ms'],4), 'frac', round(d['roofline']['frac'],3), 'norm', d['config']['last_term_norm'])"; }
for w in 32768 12000 100000; do
  echo "general"; run $w
  for b in 10 11 12 13; do echo "paced band 2^$b"; SL_PW_BAND=$b run $w; done
done
