# tools/ab_slack.sh: lead (panels) of the paced panel kernel on the headline input, interleaved
cd /root/repo
run() { python bench.py --no-sweep --no-cpu-baseline --steps 50 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('   dev ms', round(d['roofline']['launch_ms'],4))"; }
for i in 1 2; do for sl in 0 1 2 3 6; do echo "slack $sl"; SL_PW_SLACK=$sl run; done; done
