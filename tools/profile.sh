#!/bin/bash
# rocprofv3 evidence for a bench configuration (run on the GPU box via gpurun).
# usage: bash tools/profile.sh <tag> [bench args...]      e.g.  bash tools/profile.sh r02_uniform --bandwidth 0
# Passes (counters in their own runs, never combined with trace domains other than --kernel-trace):
#   stats      --kernel-trace --stats            per-kernel durations
#   pmc_fetch  --pmc FETCH_SIZE                  HBM-side read requests   (x2 on gfx950, MI355X_MICROARCH.md)
#   pmc_write  --pmc WRITE_SIZE                  HBM-side writes
#   pmc_tcc    --pmc TCC_HIT_sum TCC_MISS_sum    L2 hit rate of the gathers
TAG=${1:-r02}; shift
OUT=/root/repo/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python /root/repo/bench.py --no-cpu-baseline --no-sweep"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- $B --steps 30 --warmup 3 "$@" > $OUT/bench_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o fetch -- $B --steps 10 --warmup 2 "$@" > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o write -- $B --steps 10 --warmup 2 "$@" > $OUT/bench_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/pmc_tcc -o tcc -- $B --steps 10 --warmup 2 "$@" > $OUT/bench_tcc.log 2>&1
grep '^{"metric"' $OUT/bench_stats.log | tail -n 1 > $OUT/bench_under_rocprof.json
ls -la $OUT/*/* | head -40
# summarise ON the box (gpurun_out/ comes back only below 64 MiB; each rocpd database is ~22 MB), then drop the databases
NKW=${PROF_NKW:-"10000000 16 0"}
PROF_OUT=$OUT/summary python /root/repo/tools/prof_summary.py $OUT $TAG $NKW > $OUT/summary.log 2>&1
find $OUT -name '*.db' -delete
ls -la $OUT/summary
