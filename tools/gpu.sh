#!/bin/bash
# gpurun wrapper for THIS container: retries only when no GPU slot was free (exit code 3: nothing
# ran, nothing was charged); any other outcome is returned as it is.
#   tools/gpu.sh <timeout-seconds> '<command>'
t=$1; shift
for k in 1 2 3 4 5 6; do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 45
done
exit 3
