#!/bin/bash
# three sweep groups (3 + 3 + 2 subdomains), one per hardware queue that does not share a pipe with another: every deal (a, b)
cd "$(dirname "$0")/.." || exit 1
for a in 0 1 2 3; do for b in 0 1 2 3; do
  r=$(HPDDM_HIP_STREAMS=3 HPDDM_HIP_STREAM_PATTERN=$a,$b timeout 120 python bench.py --grid 128 --no-two-level --steps 30 --no-cpu-baseline --no-gmres 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3))")
  echo "3 groups $a,$b $r"
done; done
