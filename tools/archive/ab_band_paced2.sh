cd /root/repo
run() { python bench.py --bandwidth $1 --no-sweep --no-cpu-baseline --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('   w', d['config']['half_bandwidth'], d['roofline']['kernel'][:16], 'ms', round(d['roofline']['launch_ms'],4), 'frac', round(d['roofline']['frac'],3))"; }
for w in 32768 12000; do
  for b in 7 8 9 10; do for sl in 0 1 2 4; do echo "2^$b slack $sl"; SL_PW_BAND=$b SL_PW_SLACK=$sl run $w; done; done
done
