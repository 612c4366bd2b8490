#!/bin/bash
# where the ORDERED paced kernel's time goes (library built with EXTRA=-DSL_PWR_VARIANTS; the variants compute wrong results): SL_PW_VAR
# 2 = no LDS update, 8 = no epilogue traffic, 32 = gathers folded into 2 MB (all L2 hits, no first touches)
cd "$(dirname "$0")/.."
run() { SL_PW_VAR=$1 python bench.py --steps 30 --no-sweep --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pw var $1', round(d['roofline']['launch_ms'],4))"; }
for rep in 1 2; do for v in 0 2 8 10 32 40 42; do run $v; done; done
