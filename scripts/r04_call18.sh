#!/bin/bash
# hardware queues of the sweep groups, third pass: every deal (a, b, c) of unused streams before the streams of groups 1, 2, 3, a in {0, 1}
cd "$(dirname "$0")/.." || exit 1
for a in 0 1; do for b in 0 1 2 3; do for c in 0 1 2 3; do
  r=$(HPDDM_HIP_STREAM_PATTERN=$a,$b,$c timeout 120 python bench.py --grid 128 --no-two-level --steps 30 --no-cpu-baseline --no-gmres 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3))")
  echo "$a,$b,$c $r"
done; done; done
