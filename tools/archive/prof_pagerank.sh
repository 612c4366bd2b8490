# tools/prof_pagerank.sh: rocprofv3 --kernel-trace --stats of the PageRank demo (full solve + one query), per-kernel totals
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prpr
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prpr -o pr -- python /root/repo/tools/pagerank_query.py --thetas 1e-5 > /tmp/pr.log 2>&1
tail -c 600 /tmp/pr.log
python - <<'PY'
import sqlite3, glob
db = glob.glob("/tmp/prpr/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
rows = c.execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc limit 22").fetchall()
for r in rows: print(f"{r[0][:100]:<100} {r[1]:>7} {r[2]/1e6:>10.3f} ms {r[3]/1e3:>10.2f} us {r[4]:>6.2f} %")
PY
