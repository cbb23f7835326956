#!/bin/bash
# hardware queues of the sweep groups, second pass: groups sharing the queue of the library stream (same queue, different streams: no
# barrier between their kernels, one packet processor) against groups on queues of their own
cd "$(dirname "$0")/.." || exit 1
run() {
  timeout 300 python bench.py --grid 128 --no-two-level --steps 50 --no-cpu-baseline --no-gmres 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  configs[1]: applies/s', round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'frac', round(d['roofline']['frac'],4))"
}
echo "== 4 groups, all on the queue of the library stream (pattern 0,3,3)"; HPDDM_HIP_STREAM_PATTERN=0,3,3 run
echo "== 4 groups, (g0 g1 g2) (g3) (pattern 0,3,0)"; HPDDM_HIP_STREAM_PATTERN=0,3,0 run
echo "== 4 groups, (g0 g1) (g2 g3) (pattern 0,0,3)"; HPDDM_HIP_STREAM_PATTERN=0,0,3 run
echo "== 8 groups, all on one queue (pattern 0,3,3,3,3,3,3)"; HPDDM_HIP_STREAMS=8 HPDDM_HIP_STREAM_PATTERN=0,3,3,3,3,3,3 run
echo "== 8 groups, default creation"; HPDDM_HIP_STREAMS=8 run
echo "== 8 groups, pairs per queue (0,0,0,0,0,0,0 = round robin)"; HPDDM_HIP_STREAMS=8 HPDDM_HIP_STREAM_PATTERN=0,0,0,0,0,0,0 run
echo "== 4 groups, GPU_MAX_HW_QUEUES=1"; GPU_MAX_HW_QUEUES=1 run
echo "== 8 groups, GPU_MAX_HW_QUEUES=1"; GPU_MAX_HW_QUEUES=1 HPDDM_HIP_STREAMS=8 run
echo "== 2 groups, GPU_MAX_HW_QUEUES=1"; GPU_MAX_HW_QUEUES=1 HPDDM_HIP_STREAMS=2 run
echo "== 4 groups default (reference)"; run
