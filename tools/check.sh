mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-sweep 2>&1 | grep -E "metric|Error|error" | cut -c1-400
python bench.py > gpurun_out/bench_default.txt 2>&1; tail -1 gpurun_out/bench_default.txt | cut -c1-3000
bash tools/profile.sh r02 > /dev/null 2>&1
ls gpurun_out/prof_r02
