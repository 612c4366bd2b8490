#!/bin/bash
# order-free kernel: lead a block may take over the slowest block of its XCD (chunks of 64 entries; 1048576 = unpaced), ms per step
cd "$(dirname "$0")/.."
run() { SL_PWR_VAR=${2:-0} SL_PW_SLACK=$1 python bench.py --steps 30 --order 2 --no-sweep --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lead $1 var ${2:-0}', round(d['roofline']['launch_ms'],4), d['config']['last_term_norm'])"; }
for rep in 1 2; do
python bench.py --steps 30 --order 0 --no-sweep --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ordered', round(d['roofline']['launch_ms'],4))"
for l in 1048576 4 8 16 24 48 96 200; do run $l; done
done
