#!/bin/bash
# L2 counters of the order-free kernel's measurement variants (library built with EXTRA=-DSL_PWR_VARIANTS): where do its misses come from?
OUT=/root/repo/gpurun_out/pmc_pwr_var; mkdir -p $OUT; : > $OUT/summary.txt
cd /tmp && export TMPDIR=/tmp
B="python /root/repo/bench.py --no-cpu-baseline --no-sweep --steps 8 --warmup 2 --order 2"
for v in ${VARS:-0 4 10 32 42}; do
  for set in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCC_REQ_sum GRBM_GUI_ACTIVE"; do
    rm -rf $OUT/p
    SL_PWR_VAR=$v timeout -k 10 180 rocprofv3 --pmc $set --kernel-trace -d $OUT/p -o p -- $B "$@" > $OUT/log.txt 2>&1
    python - "$OUT/p" "$v" <<'PY' >> $OUT/summary.txt
import sqlite3, sys, glob
for db in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    for kn, cn, avg, n in sqlite3.connect(db).execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall():
        if "sl_pwr" in kn and n >= 4: print(f"var {sys.argv[2]:>3} {kn[:40]:<40} {cn:<24} {avg:14.4e}")
PY
  done
done
rm -rf $OUT/p
cat $OUT/summary.txt
