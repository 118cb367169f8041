#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy (exit code 3 = nothing charged).  usage: tools/grun.sh <timeout_s> <log> '<command>'
T=$1; LOG=$2; shift; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@" > $LOG 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
