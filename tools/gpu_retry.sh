#!/bin/bash
# Retry a gpurun call while the pod's GPU slots are busy (exit code 3 = nothing charged).  usage: tools/gpu_retry.sh TIMEOUT 'command'
T=$1; shift
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  echo "[gpu_retry] attempt $i: no slot; sleeping 90 s"
  sleep 90
done
exit 3
