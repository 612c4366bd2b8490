#!/bin/bash
# Everything round 4 wants from the GPU box, in stages that can be run one per gpurun call (each bounded by its own `timeout`):
#   bash tools/r04_gpu_session.sh tests        the whole -m gpu suite
#   bash tools/r04_gpu_session.sh micro        tools/ldsdma_bench, tools/layout_stress (item 3)
#   bash tools/r04_gpu_session.sh optin        the opt-in paths against their defaults: CG with p.Ap fused, push rounds in one workgroup
#   bash tools/r04_gpu_session.sh bench        default bench line + N = 2 / 8 self-started ranks
#   bash tools/r04_gpu_session.sh profile      rocprofv3 stats + counters of the headline (tools/profile.sh)
#   bash tools/r04_gpu_session.sh optin_fuzz [s]   tests/fuzz_campaign.py with every opt-in path on
#   bash tools/r04_gpu_session.sh fuzz [s]     tests/fuzz_dist.py, one job at a time, s seconds (default 600)
cd /root/repo
O=gpurun_out
mkdir -p $O
case "$1" in
tests)
    timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 ;;
micro)
    timeout 200 tools/ldsdma_bench 20 > $O/r04_ldsdma_bench.txt 2>&1; cat $O/r04_ldsdma_bench.txt
    timeout 100 tools/launch_floor > $O/r04_launch_floor.txt 2>&1; cat $O/r04_launch_floor.txt
    for p in 8 7; do
        timeout 300 tools/layout_stress $p 1500 400000 2 > $O/r04_layout_stress_$p.json 2> $O/r04_layout_stress_$p.err
        cat $O/r04_layout_stress_$p.json; echo "misfits reported: $(grep -c 'did not fit' $O/r04_layout_stress_$p.err)"; head -c 1200 $O/r04_layout_stress_$p.err
    done ;;
optin)
    for f in 0 1; do SL_CG_FUSED_DOT=$f timeout 300 python tools/cg_bench.py > $O/r04_cg_bench_fused$f.json 2>$O/r04_cg_fused$f.err; cat $O/r04_cg_bench_fused$f.json; done
    SL_CG_FUSED_DOT=1 timeout 600 python -m pytest tests/test_gpu_cg.py -q 2>&1 | tail -3
    timeout 900 python -m pytest tests/test_gpu_session.py -q -k "small_rounds or wide_batch" 2>&1 | tail -3
    for f in 0 1; do SL_PUSH_SMALL=$f timeout 600 python tools/pagerank_query.py --no-full-solve > $O/r04_pagerank_small$f.json 2>$O/r04_pagerank_small$f.err; tail -c 2500 $O/r04_pagerank_small$f.json; echo; done
    timeout 900 python -m pytest tests/test_gpu_pagerank.py -q -k index_only 2>&1 | tail -3
    for f in 0 1; do SL_PW_INDEX_ONLY=$f timeout 900 python tools/pagerank_query.py --thetas 1e-5 > $O/r04_pagerank_idx$f.json 2>$O/r04_pagerank_idx$f.err; head -c 900 $O/r04_pagerank_idx$f.json; echo; done
    for w in 8 16 32; do SL_QUERY_WIDE=$w timeout 600 python tools/pagerank_query.py --no-full-solve > $O/r04_pagerank_wide$w.json 2>$O/r04_pagerank_wide$w.err; tail -c 1500 $O/r04_pagerank_wide$w.json; echo; done ;;
bench)
    timeout 900 python bench.py > $O/r04_bench_default.json 2>$O/r04_bench_default.err; cat $O/r04_bench_default.json | cut -c1-1500
    timeout 600 python bench.py --gpus 2 --steps 20 > $O/r04_bench_2ranks_1gpu.json 2>$O/r04_bench_2ranks.err; cut -c1-600 $O/r04_bench_2ranks_1gpu.json ;;
profile)
    bash tools/profile.sh r04_uniform --bandwidth 0 2>&1 | tail -12 ;;
optin_fuzz)      # the random parity campaign with every opt-in path of the round switched on: whatever runs must still equal the oracle
    SL_PUSH_SMALL=1 SL_QUERY_WIDE=4 SL_CG_FUSED_DOT=1 SL_PW_INDEX_ONLY=1 timeout $(( ${2:-240} + 120 )) python tests/fuzz_campaign.py --seconds ${2:-240} > $O/r04_fuzz_optin.json 2> $O/r04_fuzz_optin.err; tail -c 1200 $O/r04_fuzz_optin.json ;;
fuzz)
    timeout $(( ${2:-600} + 120 )) python tests/fuzz_dist.py --seconds ${2:-600} --min-world 5 --seed0 ${3:-4410} > $O/r04_fuzz_dist_seq_${3:-4410}.json 2> $O/r04_fuzz_dist_seq.err; tail -c 800 $O/r04_fuzz_dist_seq_${3:-4410}.json ;;
*)
    echo "usage: $0 tests|micro|optin|bench|profile|fuzz [seconds] [seed]"; exit 2 ;;
esac
