# tools/ab_prev_uniform.sh: interleaved A/B of the built library against build/libsublinear_hip_prev.so on bench.py's uniform-column input
cd /root/repo
run() { python bench.py "$@" --no-sweep --no-cpu-baseline --steps 50 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('   dev ms', round(d['roofline']['launch_ms'],4))"; }
for i in 1 2 3; do
  echo "prev uniform"; SUBLINEAR_HIP_LIB=build/libsublinear_hip_prev.so run
  echo "head uniform"; run
done
