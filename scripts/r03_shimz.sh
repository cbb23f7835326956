#!/bin/bash
mkdir -p gpurun_out/r03
export OMP_NUM_THREADS=8
timeout 600 python -m pytest tests/test_gpu_dropin.py -q -m gpu -x -k "c_example or c_api" > gpurun_out/r03/shimz_tests.log 2>&1; echo "shim tests rc=$?"; tail -5 gpurun_out/r03/shimz_tests.log
timeout 400 python bench.py --no-cpu-baseline --no-configs-1 --no-geneo --no-shares --steps 5 > gpurun_out/r03/setup_breakdown.log 2>&1; grep '^{"metric"' gpurun_out/r03/setup_breakdown.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['config']['setup_seconds'], d['config']['setup_seconds_by_phase_summed_over_subdomains'], d['roofline']['frac'])"
