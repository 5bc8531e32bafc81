#!/bin/bash
# keep asking for a GPU slot until one is free (exit 3 = nothing charged): tools/gpurun_retry.sh <timeout-s> '<command>'
t=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
