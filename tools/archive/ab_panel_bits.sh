# tools/ab_panel_bits.sh: panel width of the column-panel layouts (2^15 / 2^16 / 2^17 columns; variant builds under build/), headline input, leads 1 and 2
cd /root/repo
run() { python bench.py --no-sweep --no-cpu-baseline --steps 50 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('   dev ms', round(d['roofline']['launch_ms'],4))"; }
for i in 1 2; do
  for sl in 1 2; do
    echo "2^15 lead $sl"; SL_PW_SLACK=$sl SUBLINEAR_HIP_LIB=build/lib_pb15.so run
    echo "2^16 lead $sl"; SL_PW_SLACK=$sl run
    echo "2^17 lead $sl"; SL_PW_SLACK=$sl SUBLINEAR_HIP_LIB=build/lib_pb17.so run
  done
done
