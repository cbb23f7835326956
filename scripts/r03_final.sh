#!/bin/bash
# end of round 3: the whole GPU suite, then the default bench line with the final library
mkdir -p gpurun_out/r03
export OMP_NUM_THREADS=8
timeout 1200 python -m pytest tests -q -m gpu > gpurun_out/r03/gpu_tests_final.log 2>&1; echo "gpu tests rc=$?"; tail -4 gpurun_out/r03/gpu_tests_final.log | cut -c1-300
( time timeout 900 python bench.py ) > gpurun_out/r03/bench_default_final_stdout.log 2> gpurun_out/r03/bench_default_final_stderr.log
grep '^{"metric"' gpurun_out/r03/bench_default_final_stdout.log | tail -1 > gpurun_out/r03/bench_default_final_stdout.json
python -c "
import json; d=json.load(open('gpurun_out/r03/bench_default_final_stdout.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic_source'), d['phases_ms'], d['two_level']['deflation_ms'], d['two_level'].get('deflation_mfma_mu8'))"
tail -3 gpurun_out/r03/bench_default_final_stderr.log
