# usage: prof_pagerank_lib.sh <lib or ""> : per-kernel totals of the PageRank demo with an alternative library build
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prpr
[ -n "$1" ] && export SUBLINEAR_HIP_LIB=/root/repo/$1
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prpr -o pr -- python /root/repo/tools/pagerank_query.py --thetas 1e-7 > /tmp/pr.log 2>&1
grep "^{" /tmp/pr.log | python -c "
import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('full', d['full_solve']['device_ms'], [round(q['device_ms'],2) for q in d['queries']])"
python - <<'PY'
import sqlite3, glob
db = glob.glob("/tmp/prpr/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
rows = c.execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc limit 4").fetchall()
for r in rows: print(f"{r[0][:70]:<70} {r[1]:>7} {r[2]/1e6:>10.3f} s  {r[3]/1e3:>10.2f} ms {r[4]:>6.2f} %")
PY
