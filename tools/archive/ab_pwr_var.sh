#!/bin/bash
# Where the order-free kernel's time goes (library built with EXTRA=-DSL_PWR_VARIANTS): SL_PWR_VAR 1 = plain read-add-write instead of
# LDS atomics, 2 = no LDS update, 4 = no gathers, 8 = no epilogue traffic (sums WRONG in all of them: timing only), 16 = L2 prefetch of
# the vector lines SL_PW_SLACK chunk positions ahead by return-less atomics
cd "$(dirname "$0")/.."
run() { SL_PWR_VAR=$1 SL_PW_SLACK=${2:-0} python bench.py --steps 30 --order ${3:-2} --no-sweep --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('order ${3:-2} var $1 lead ${2:-0}', round(d['roofline']['launch_ms'],4), d['config']['last_term_norm'])"; }
for rep in 1 2 3; do
run 0 0 0; run 0; run 1; run 2; run 4; run 6; run 10; run 14
done
