# tools/ab_prev.sh: interleaved A/B of the built library against build/libsublinear_hip_prev.so (an older build) on bench.py's banded and uniform inputs
cd /root/repo
run() { python bench.py "$@" --no-sweep --no-cpu-baseline --steps 50 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('   dev ms', round(d['roofline']['launch_ms'],4))"; }
for i in 1 2 3; do
  echo "prev w4096"; SUBLINEAR_HIP_LIB=build/libsublinear_hip_prev.so run --bandwidth 4096
  echo "head w4096"; run --bandwidth 4096
  echo "prev w512"; SUBLINEAR_HIP_LIB=build/libsublinear_hip_prev.so run --bandwidth 512
  echo "head w512"; run --bandwidth 512
done
echo "prev uniform"; SUBLINEAR_HIP_LIB=build/libsublinear_hip_prev.so run
echo "head uniform"; run
