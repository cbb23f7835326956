#!/bin/bash
# streams of the device levels x hardware queues of the runtime: the device levels of a 129^3 Cholesky factorisation, level by level
cd "$(dirname "$0")/.." || exit 1
for combo in "4 0" "8 0" "8 8" "16 16" "16 0"; do
  set -- $combo
  echo "== streams $1 GPU_MAX_HW_QUEUES ${2/#0/default}"
  if [ "$2" != "0" ]; then export GPU_MAX_HW_QUEUES=$2; else unset GPU_MAX_HW_QUEUES; fi
  HPDDM_HIP_FACTOR_STREAMS=$1 HPDDM_HIP_PROFILE=1 timeout 300 python scripts/time_numfact.py 129 chol device 2>&1 | grep "device level\|N=129" | tail -17 | awk '/device level [0-9]+:/ {printf "%s%s ", $4, $11} /levels|N=129/ {print ""; print}'
done
