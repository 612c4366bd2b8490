# wide bands on the GPU box: multi-pass window kernel (default) against the general kernel (SL_MPASS=0)
cd /root/repo
run() { python bench.py --bandwidth $1 --no-sweep --no-cpu-baseline --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('w', d['config']['half_bandwidth'], 'ms', round(d['roofline']['launch_ms'],4), 'frac', round(d['roofline']['frac'],3))"; }
for w in 12000 32768 100000; do echo "mpass w=$w"; run $w; echo "general w=$w"; SL_MPASS=0 run $w; done
echo "stencil mpass"; python tools/stencil_bench.py 2>/dev/null | tail -2
echo "stencil general"; SL_MPASS=0 python tools/stencil_bench.py 2>/dev/null | tail -2
