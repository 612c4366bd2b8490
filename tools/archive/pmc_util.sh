#!/bin/bash
# utilisation counters of the step kernel (separate rocprofv3 --pmc passes, --kernel-trace only): who is busy, who waits, where the
# reads are served.  usage: bash tools/pmc_util.sh <tag> [bench args...]   -> gpurun_out/pmc_<tag>/summary.txt
TAG=${1:-r02_uniform_util}; shift
OUT=/root/repo/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python /root/repo/bench.py --no-cpu-baseline --no-sweep --steps 8 --warmup 2"
i=0
# (a pass with TA_ADDR_STALLED_BY_TC_CYCLES_sum / TA_DATA_STALLED_BY_TC_CYCLES_sum / TA_TA_BUSY_sum aborted inside rocprofv3 and hung
# until the box's limit: not collected; every pass now runs under its own timeout)
for set in "TA_BUSY_avr TCC_BUSY_avr GRBM_TA_BUSY GRBM_GUI_ACTIVE" \
           "TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" \
           "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
           "SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o p -- $B "$@" > $OUT/p$i.log 2>&1 || echo "pass $i ($set) failed or timed out" >> $OUT/failed.txt
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
            print(f"{kn[:60]:<60} {cn:<38} {avg:16.1f}  ({n} dispatches)")
PY
find $OUT -name '*.db' -delete
cat $OUT/summary.txt
