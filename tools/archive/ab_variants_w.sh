# tools/ab_variants_w.sh <bandwidth> <variants...>: as ab_variants.sh, for a given column structure
cd /root/repo
w=$1; shift
run() { python bench.py --bandwidth $w --no-sweep --no-cpu-baseline --steps 50 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('   dev ms', round(d['roofline']['launch_ms'],4))"; }
for i in 1 2 3; do
  echo head; run
  for v in "$@"; do echo $v; SUBLINEAR_HIP_LIB=build/lib_$v.so run; done
done
