# per-dispatch durations of sl_pw_kernel in the PageRank demo, split by LDS size (main stream / long rows' stream)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prpr
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prpr -o pr -- python /root/repo/tools/pagerank_query.py --thetas 1e-7 > /tmp/pr.log 2>&1
python - <<'PY'
import sqlite3, glob
db = glob.glob("/tmp/prpr/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
print(cols)
q = "select name, lds_size, grid_x, count(*), avg(duration), min(duration), max(duration) from kernels where name like '%sl_pw_kernel%' or name like '%sl_long_rows%' group by name, lds_size, grid_x order by 1,2"
try:
    for r in c.execute(q): print(r)
except Exception as e:
    print("err", e)
PY
