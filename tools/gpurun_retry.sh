#!/bin/bash
# gpurun with a retry ONLY for "no box / slot free right now" (exit code 3: nothing ran, nothing charged):
#   bash tools/gpurun_retry.sh <timeout_s> '<command>'
t=$1; shift
for i in 1 2 3 4 5 6 7 8; do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  echo "[gpurun_retry] no slot (attempt $i), sleeping 120 s"; sleep 120
done
exit 3
