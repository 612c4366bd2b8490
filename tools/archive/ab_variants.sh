# tools/ab_variants.sh: variant builds under build/lib_<name>.so against the built library on the headline input, interleaved
cd /root/repo
run() { python bench.py --no-sweep --no-cpu-baseline --steps 50 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('   dev ms', round(d['roofline']['launch_ms'],4))"; }
for i in 1 2; do
  echo head; run
  for v in "$@"; do echo $v; SUBLINEAR_HIP_LIB=build/lib_$v.so run; done
done
