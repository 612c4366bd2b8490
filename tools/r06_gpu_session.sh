#!/bin/bash
# Round 6's stages on the GPU box, ONE stage per gpurun call (tools/gpucall.sh logs each to profiles/r06_gpu_runs.txt), each bounded by `timeout`.
# ORDER: evidence for HEAD first — tests, bench, profile — then microbenchmarks and the opt-in A/Bs.
#   tests_core   every single-process -m gpu file, core parity first (tests/conftest.py orders them), per-test durations kept
#   tests_multi  the multi-process files (partitioned, dist_abi, multi_device, bench)
#   tests        the whole suite exactly as the driver runs it (-x)
#   bench        the default bench line -> gpurun_out/r06_bench_default.json
#   profile      rocprofv3 stats + counters of the headline (tools/profile.sh) -> gpurun_out/prof_r06_uniform/
#   micro        tools/split_bench, tools/ldsdma_bench, tools/launch_floor
#   optin        every opt-in path against its default
# At most 8 processes on the device, one job at a time.
cd /root/repo
O=gpurun_out
mkdir -p $O
MULTI="tests/test_gpu_partitioned.py tests/test_gpu_dist_abi.py tests/test_gpu_multi_device.py tests/test_gpu_bench.py"
case "$1" in
tests_core)
    IGN=""; for f in $MULTI; do IGN="$IGN --ignore=$f"; done
    timeout 2400 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider --durations=40 $IGN 2>&1 | tee $O/r06_pytest_gpu_core.txt | tail -60 ;;
tests_multi)
    timeout 2400 python -m pytest $MULTI -q -m gpu --timeout 900 -p no:cacheprovider --durations=25 2>&1 | tee $O/r06_pytest_gpu_multi.txt | tail -45 ;;
tests)
    timeout 2400 python -m pytest tests -x -q -m gpu --timeout 900 -p no:cacheprovider --durations=40 2>&1 | tee $O/r06_pytest_gpu.txt | tail -60 ;;
bench)
    timeout 900 python bench.py > $O/r06_bench_default.json 2>$O/r06_bench_default.err; cut -c1-3000 $O/r06_bench_default.json; tail -n 5 $O/r06_bench_default.err
    # the band kernel's pipeline was rebuilt this round: the banded variant of the recipe (round 3: 0.326 ms = 0.90) and the 16-wave window (0.81)
    for w in 4096 8192; do timeout 600 python bench.py --bandwidth $w --no-cpu-baseline --no-sweep > $O/r06_bench_w$w.json 2>$O/r06_bench_w$w.err; cut -c1-1200 $O/r06_bench_w$w.json; done ;;
profile)
    bash tools/profile.sh r06_uniform --bandwidth 0 2>&1 | tail -12 ;;
micro)
    timeout 300 tools/split_bench 10 > $O/r06_split_bench.txt 2>&1; cat $O/r06_split_bench.txt
    timeout 200 tools/ldsdma_bench 20 > $O/r06_ldsdma_bench.txt 2>&1; cat $O/r06_ldsdma_bench.txt
    timeout 100 tools/launch_floor > $O/r06_launch_floor.txt 2>&1; cat $O/r06_launch_floor.txt ;;
optin)
    for f in 0 1; do SL_CG_FUSED_DOT=$f timeout 300 python tools/cg_bench.py > $O/r06_cg_bench_fused$f.json 2>$O/r06_cg_fused$f.err; cat $O/r06_cg_bench_fused$f.json; done
    for f in 0 1; do SL_PUSH_SMALL=$f timeout 600 python tools/pagerank_query.py --no-full-solve > $O/r06_pagerank_small$f.json 2>$O/r06_pagerank_small$f.err; tail -c 2500 $O/r06_pagerank_small$f.json; echo; done
    for f in 0 1; do SL_PW_INDEX_ONLY=$f timeout 900 python tools/pagerank_query.py --thetas 1e-5 > $O/r06_pagerank_idx$f.json 2>$O/r06_pagerank_idx$f.err; head -c 900 $O/r06_pagerank_idx$f.json; echo; done
    timeout 900 bash tools/ab_epi_rows.sh > $O/r06_ab_epi_rows.txt 2>&1; cat $O/r06_ab_epi_rows.txt
    timeout 600 python tools/walk_bench.py > $O/r06_walk_bench.json 2>$O/r06_walk_bench.err; tail -c 1200 $O/r06_walk_bench.json; echo
    for w in 8 16 32; do SL_QUERY_WIDE=$w timeout 600 python tools/pagerank_query.py --no-full-solve > $O/r06_pagerank_wide$w.json 2>$O/r06_pagerank_wide$w.err; tail -c 1500 $O/r06_pagerank_wide$w.json; echo; done ;;
*)
    echo "usage: $0 tests_core|tests_multi|tests|bench|profile|micro|optin"; exit 2 ;;
esac
