#!/bin/bash
# L1 (TCP) / translation / L2 queue counters of the step kernel (separate rocprofv3 --pmc passes, --kernel-trace only; every pass under
# its own timeout).  usage: bash tools/pmc_tcp.sh <tag> [bench args...]   -> gpurun_out/pmc_<tag>/summary.txt
TAG=${1:-r02_uniform_tcp}; shift
OUT=/root/repo/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python /root/repo/bench.py --no-cpu-baseline --no-sweep --steps 8 --warmup 2"
i=0
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCC_TAG_STALL_sum TCC_STREAMING_REQ_sum TCC_NC_REQ_sum" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum TCC_EA0_RDREQ_IO_CREDIT_STALL_sum" \
           "TD_TC_STALL_sum TD_TD_BUSY_sum TD_LOAD_WAVEFRONT_sum"; do
  i=$((i+1))
  timeout -k 10 180 rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o p -- $B "$@" > $OUT/p$i.log 2>&1 || echo "pass $i ($set) failed or timed out" >> $OUT/failed.txt
done
python - "$OUT" <<'PY' > $OUT/summary.txt
import sqlite3, sys, glob
out = sys.argv[1]
print("# per launch of the step kernel (avg over dispatches); counters from separate passes")
for db in sorted(glob.glob(out + "/p*/*/*.db") + glob.glob(out + "/p*/*.db")):
    try:
        rows = sqlite3.connect(db).execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
    except Exception as e:
        print(db, e); continue
    for kn, cn, avg, n in rows:
        if any(t in kn for t in ("sl_pw_kernel", "sl_band_kernel", "sl_rows_kernel", "sl_panel_kernel")) and n >= 4:
            print(f"{kn[:60]:<60} {cn:<44} {avg:16.1f}  ({n} dispatches)")
PY
find $OUT -name '*.db' -delete
cat $OUT/summary.txt; cat $OUT/failed.txt 2>/dev/null
