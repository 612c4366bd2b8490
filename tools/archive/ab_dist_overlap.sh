# tools/ab_dist_overlap.sh: the partitioned step of the C ABI with the exchange AFTER the whole step (SL_DIST_OVERLAP=0) against the
# boundary-first form (edge blocks, halo ticket and pulls on the side stream beside the interior).  One GPU only here: world size 1
# with the form forced on (=2: same launches and stream topology, no peers), and two ranks SHARING the GPU (each 5 * 10^6 rows).
cd /root/repo
line() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('  ms/step', round(d['ms_per_step'],4), d['config']['exchange'][:60])"; }
one() { SL_BENCH_FORCE_ABI=1 SL_DIST_OVERLAP=$1 python bench.py --bandwidth 4096 --no-sweep --no-cpu-baseline --steps 50 2>/dev/null | line; }
two() {
  SL_DIST_OVERLAP=$1 SL_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) \
    bench.py --gpus 2 --rows 5000000 --k 16 --bandwidth 4096 --steps 50 --warmup 5 --no-cpu-baseline --no-sweep 2>/dev/null | line; }
for i in 1 2; do
  echo "world 1, exchange after the step"; one 0
  echo "world 1, boundary-first (forced)"; one 2
  echo "2 ranks on one GPU, exchange after the step"; two 0
  echo "2 ranks on one GPU, boundary-first"; two 1
done
