#!/bin/bash
# One logged gpurun call: the tail gpurun prints lands in gpurun_out/r06_call_<tag>.txt and one line
# (time, tag, box seconds, exit code, command) is appended to profiles/r06_gpu_runs.txt when the call ENDS, not at the end of the round.
# usage: tools/gpucall.sh <tag> <timeout-seconds> '<command run on the GPU box from the repo root>'
# ONE call at a time; never more than 8 processes on the device, never two jobs side by side (round 4 lost a box and the pool that way).
TAG=$1; TMO=$2; shift 2
cd /root/repo; mkdir -p gpurun_out
LOG=gpurun_out/r06_call_$TAG.txt
T0=$(date +%s)
/usr/local/graft/bin/gpurun --timeout $TMO -- "$@" > $LOG 2>&1
RC=$?
T1=$(date +%s)
printf '%s  call %-22s  wall %5ds  gpurun rc %d  head %s  cmd: %s\n' "$(date -u +%H:%M:%S)" "$TAG" $((T1-T0)) $RC "$(git rev-parse --short HEAD)" "$*" >> profiles/r06_gpu_runs.txt
tail -n 40 $LOG
exit $RC
