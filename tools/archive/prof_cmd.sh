#!/bin/bash
# usage: tools/prof_cmd.sh <tag> <command...> : rocprofv3 --kernel-trace --stats of any command of this repo; per-kernel totals
# (top_kernels view of the rocpd database) to stdout and to gpurun_out/<tag>_kernel_stats.txt.  Run from the repo root on the GPU box.
tag=$1; shift
R=$(pwd); mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_$tag
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o p -- "$@" > /tmp/prof_$tag.log 2>&1 < /dev/null
grep -v "^W2\|^E2\|^I2\|amdgpu.ids" /tmp/prof_$tag.log | tail -5
python3 - $tag > $R/gpurun_out/${tag}_kernel_stats.txt <<'PY'
import sqlite3, glob, sys
dbs = glob.glob(f"/tmp/prof_{sys.argv[1]}/**/*.db", recursive=True)
if not dbs:
    print("no database written"); sys.exit(0)
c = sqlite3.connect(dbs[0])
print(f"{'kernel':<90} {'calls':>7} {'total_ms':>10} {'avg_us':>10} {'pct':>7}")
for name, calls, tot, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc"):
    print(f"{name[:90]:<90} {calls:>7} {tot / 1e3:>10.3f} {avg:>10.2f} {pct:>7.2f}")
PY
cat $R/gpurun_out/${tag}_kernel_stats.txt
