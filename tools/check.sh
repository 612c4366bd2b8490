mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -3
python bench.py > gpurun_out/bench_default.txt 2>&1; tail -1 gpurun_out/bench_default.txt | cut -c1-2500
rm -rf gpurun_out/prof_r02; bash tools/profile.sh r02 > /dev/null 2>&1; ls gpurun_out/prof_r02
