#!/bin/bash
# L2 / L1 counters of the step kernel for two ways of running the same instance, side by side (separate rocprofv3 --pmc passes,
# --kernel-trace only, each under its own timeout).  usage: tools/pmc_pair.sh <tag> "<bench args A>" "<bench args B>"
TAG=${1:-pair}; A=${2:---order 0}; Bv=${3:---order 2}
OUT=/root/repo/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python /root/repo/bench.py --no-cpu-baseline --no-sweep --steps 8 --warmup 2"
for which in A B; do
  if [ $which = A ]; then ARGS=$A; else ARGS=$Bv; fi
  i=0
  for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum" \
             "TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum" \
             "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_128B_sum" \
             "TA_BUSY_avr TCC_BUSY_avr GRBM_GUI_ACTIVE" \
             "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD"; do
    i=$((i+1))
    timeout -k 10 180 rocprofv3 --pmc $set --kernel-trace -d $OUT/$which$i -o p -- $B $ARGS > $OUT/$which$i.log 2>&1 || echo "pass $which$i ($set) failed or timed out" >> $OUT/failed.txt
  done
done
python - "$OUT" "$A" "$Bv" <<'PY' > $OUT/summary.txt
import sqlite3, sys, glob
out, A, B = sys.argv[1:4]
print(f"# per launch of the step kernel (avg over dispatches); counters from separate passes.  A = bench.py {A}   B = bench.py {B}")
for db in sorted(glob.glob(out + "/[AB]*/*/*.db") + glob.glob(out + "/[AB]*/*.db")):
    which = db[len(out) + 1]
    try:
        rows = sqlite3.connect(db).execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
    except Exception as e:
        print(db, e); continue
    for kn, cn, avg, n in rows:
        if any(t in kn for t in ("sl_pw", "sl_band_kernel", "sl_rows_kernel", "sl_panel_kernel")) and n >= 4:
            print(f"{which} {kn[:44]:<44} {cn:<34} {avg:16.1f}  ({n} dispatches)")
PY
find $OUT -name '*.db' -delete
cat $OUT/summary.txt
