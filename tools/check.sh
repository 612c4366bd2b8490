mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -25
timeout 600 python tools/pagerank_query.py --n 10000000 > gpurun_out/pagerank_10m.json 2> gpurun_out/pagerank_10m.err; tail -c 3000 gpurun_out/pagerank_10m.json; tail -5 gpurun_out/pagerank_10m.err
