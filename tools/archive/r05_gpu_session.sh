#!/bin/bash
# Everything round 5 wants from the GPU box, ONE stage per gpurun call (tools/gpucall.sh logs each), each bounded by its own `timeout`.
# ORDER: evidence for HEAD first — tests, bench, profile — and only then the opt-in A/Bs and the microbenchmarks.
#   bash tools/r05_gpu_session.sh tests        the whole -m gpu suite, log kept in gpurun_out/r05_pytest_gpu.txt
#   bash tools/r05_gpu_session.sh bench        the default bench line (parity_gate, roofline, cpu_baseline) -> gpurun_out/r05_bench_default.json
#   bash tools/r05_gpu_session.sh profile      rocprofv3 stats + counters of the headline (tools/profile.sh) -> gpurun_out/prof_r05_uniform/
#   bash tools/r05_gpu_session.sh optin        every opt-in path against its default, at size (CG fused dot, small rounds, wide batches, index-only stream)
#   bash tools/r05_gpu_session.sh micro        tools/split_bench (stream warmed by other CUs of the XCD?), tools/ldsdma_bench, tools/launch_floor
#   bash tools/r05_gpu_session.sh optin_fuzz [s]   tests/fuzz_campaign.py with every opt-in path on
# No many-process campaign here: at most 8 processes on the device, one job at a time (round 4 lost a box and the pool to 48).
cd /root/repo
O=gpurun_out
mkdir -p $O
case "$1" in
tests)
    timeout 2000 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | tee $O/r05_pytest_gpu.txt | tail -25 ;;
bench)
    timeout 900 python bench.py > $O/r05_bench_default.json 2>$O/r05_bench_default.err; cut -c1-2500 $O/r05_bench_default.json; tail -n 5 $O/r05_bench_default.err ;;
profile)
    bash tools/profile.sh r05_uniform --bandwidth 0 2>&1 | tail -12 ;;
optin)
    for f in 0 1; do SL_CG_FUSED_DOT=$f timeout 300 python tools/cg_bench.py > $O/r05_cg_bench_fused$f.json 2>$O/r05_cg_fused$f.err; cat $O/r05_cg_bench_fused$f.json; done
    SL_CG_FUSED_DOT=1 timeout 600 python -m pytest tests/test_gpu_cg.py -q 2>&1 | tail -3
    timeout 900 python -m pytest tests/test_gpu_session.py tests/test_gpu_optin_oracle.py -q 2>&1 | tail -3
    for f in 0 1; do SL_PUSH_SMALL=$f timeout 600 python tools/pagerank_query.py --no-full-solve > $O/r05_pagerank_small$f.json 2>$O/r05_pagerank_small$f.err; tail -c 2500 $O/r05_pagerank_small$f.json; echo; done
    for f in 0 1; do SL_PW_INDEX_ONLY=$f timeout 900 python tools/pagerank_query.py --thetas 1e-5 > $O/r05_pagerank_idx$f.json 2>$O/r05_pagerank_idx$f.err; head -c 900 $O/r05_pagerank_idx$f.json; echo; done
    for w in 8 16 32; do SL_QUERY_WIDE=$w timeout 600 python tools/pagerank_query.py --no-full-solve > $O/r05_pagerank_wide$w.json 2>$O/r05_pagerank_wide$w.err; tail -c 1500 $O/r05_pagerank_wide$w.json; echo; done ;;
micro)
    timeout 300 tools/split_bench 10 > $O/r05_split_bench.txt 2>&1; cat $O/r05_split_bench.txt
    timeout 200 tools/ldsdma_bench 20 > $O/r05_ldsdma_bench.txt 2>&1; cat $O/r05_ldsdma_bench.txt
    timeout 100 tools/launch_floor > $O/r05_launch_floor.txt 2>&1; cat $O/r05_launch_floor.txt ;;
optin_fuzz)      # the random parity campaign with every opt-in path of the round switched on: whatever runs must still equal the oracle
    SL_PUSH_SMALL=1 SL_QUERY_WIDE=4 SL_CG_FUSED_DOT=1 SL_PW_INDEX_ONLY=1 timeout $(( ${2:-240} + 120 )) python tests/fuzz_campaign.py --seconds ${2:-240} > $O/r05_fuzz_optin.json 2> $O/r05_fuzz_optin.err; tail -c 1200 $O/r05_fuzz_optin.json ;;
*)
    echo "usage: $0 tests|bench|profile|optin|micro|optin_fuzz [seconds]"; exit 2 ;;
esac
