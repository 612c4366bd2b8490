#!/bin/bash
# rocprofv3 evidence for the headline bench configuration (run on the GPU box via gpurun).
# usage: bash tools/profile.sh <tag> [bench args...]
TAG=${1:-r01}; shift
OUT=/root/repo/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python /root/repo/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-sweep "$@" > $OUT/bench_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o fetch -- python /root/repo/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-sweep "$@" > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o write -- python /root/repo/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-sweep "$@" > $OUT/bench_write.log 2>&1
find $OUT -type f | head -50
ls -la $OUT/*/* | head -40
