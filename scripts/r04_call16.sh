#!/bin/bash
# which hardware queues the streams of the sweep groups land on (creation order): configs[1] (8 subdomains of 65^3, one-level apply)
cd "$(dirname "$0")/.." || exit 1
run() {
  timeout 300 python bench.py --grid 128 --no-two-level --steps 50 --no-cpu-baseline --no-gmres 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  configs[1]: applies/s', round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'frac', round(d['roofline']['frac'],4))"
}
for sk in 0 1 2 3; do echo "== skew $sk"; HPDDM_HIP_STREAM_SKEW=$sk run; done
echo "== skew 0, GPU_MAX_HW_QUEUES=8"; GPU_MAX_HW_QUEUES=8 run
echo "== skew 0, GPU_MAX_HW_QUEUES=2"; GPU_MAX_HW_QUEUES=2 run
echo "== 3 groups"; HPDDM_HIP_STREAMS=3 run
echo "== 2 groups"; HPDDM_HIP_STREAMS=2 run
echo "== 2 groups skew 3"; HPDDM_HIP_STREAMS=2 HPDDM_HIP_STREAM_SKEW=3 run
