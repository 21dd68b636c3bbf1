#!/bin/bash
# tools/gpurun_retry.sh TIMEOUT 'command' : gpurun, retried every two minutes while the pod's GPU slots are busy (exit code 3)
T=$1; shift
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 120
done
exit 3
