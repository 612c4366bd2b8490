# short / ragged rows on the GPU box: the fused step at 5..15 entries per row, banded; and the 2-D 5-point stencil
cd /root/repo
run() { python bench.py --k $1 --bandwidth $2 --no-sweep --no-cpu-baseline --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('k', d['config']['nnz_per_row'], 'w', d['config']['half_bandwidth'], 'ms', round(d['roofline']['launch_ms'],4), 'frac', round(d['roofline']['frac'],3))"; }
for k in 8 5 9 12 15; do for w in 512 4096; do run $k $w; done; done
python tools/stencil_bench.py --nx 3162 --ny 3162 --nz 1 2>/dev/null | tail -1
