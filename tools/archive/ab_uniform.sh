# A/B of the paced column-panel kernel on the GPU box: bash tools/ab_uniform.sh   (SL_PW_SLACK: panels of lead; 1048576 = unpaced)
cd /root/repo
run() { python bench.py --bandwidth 0 --no-sweep --no-cpu-baseline --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['launch_ms'])"; }
for s in 0 2 4 8 1048576 0 4; do echo "slack $s"; SL_PW_SLACK=$s run; done
echo dynamic; SL_COLUMN_PANELS=3 run
