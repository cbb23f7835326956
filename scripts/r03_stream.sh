#!/bin/bash
mkdir -p gpurun_out/r03
export OMP_NUM_THREADS=8
for pairs in 0 1; do
timeout 300 python bench.py --options="-hpddm_hip_deflation_pairs $pairs" --no-cpu-baseline --no-configs-1 --no-shares --no-geneo --steps 20 > gpurun_out/r03/stream_bench_$pairs.log 2>&1
grep '^{"metric"' gpurun_out/r03/stream_bench_$pairs.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('pairs=$pairs', round(d['ms_per_step'],3), d['phases_ms'], d['two_level']['deflation_ms'], d['two_level'].get('deflation_panel_GBps'))"
done
