#!/bin/bash
# developer aid: sweep the plan-builder knobs of the SpTRSV on the default bench workload (one GPU box call).
# usage: scripts/sweep_sptrsv.sh "ENV1=a ENV2=b" "ENV1=c" ...
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-two-level --no-gmres $BENCH_EXTRA 2>&1 | tail -1 |
    python -c 'import sys, json; d = json.loads(sys.stdin.readline()); print("ms/apply %.3f  sptrsv %.3f ms  frac %.3f  launches %d" % (d["ms_per_step"], d["phases_ms"]["sptrsv"], d["roofline"]["frac"], d["config"]["launches_per_sptrsv"]))'
done 2>&1 | tee -a gpurun_out/sweep.log
