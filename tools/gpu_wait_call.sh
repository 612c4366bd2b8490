#!/bin/bash
# Retry ONE gpurun call until the pool accepts it (rc 2 = refused / closed, rc 3 = no box: nothing charged either way), then log it like
# tools/gpucall.sh does.  usage: tools/gpu_wait_call.sh <tag> <timeout-seconds> <retry-every-seconds> <max-tries> '<command>'
TAG=$1; TMO=$2; EVERY=$3; MAX=$4; shift 4
cd /root/repo; mkdir -p gpurun_out
LOG=gpurun_out/r06_call_$TAG.txt
N=0
while [ $N -lt $MAX ]; do
  N=$((N+1)); T0=$(date +%s)
  /usr/local/graft/bin/gpurun --timeout $TMO -- "$@" > $LOG 2>&1; RC=$?
  T1=$(date +%s)
  if [ $RC -ne 2 ] && [ $RC -ne 3 ]; then break; fi
  if grep -q "another call" $LOG; then sleep 60; N=$((N-1)); continue; fi
  sleep $EVERY
done
printf '%s  call %-22s  wall %5ds  gpurun rc %d  head %s  refused %d times before  cmd: %s\n' "$(date -u +%H:%M:%S)" "$TAG" $((T1-T0)) $RC "$(git rev-parse --short HEAD)" $((N-1)) "$*" >> profiles/r06_gpu_runs.txt
tail -n 60 $LOG
exit $RC
