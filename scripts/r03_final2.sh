#!/bin/bash
mkdir -p gpurun_out/r03
export OMP_NUM_THREADS=8
( time timeout 600 python bench.py ) > gpurun_out/r03/bench_default_final2_stdout.log 2> gpurun_out/r03/bench_default_final2_stderr.log
grep '^{"metric"' gpurun_out/r03/bench_default_final2_stdout.log | tail -1 > gpurun_out/r03/bench_default_final2_stdout.json
python -c "
import json; d=json.load(open('gpurun_out/r03/bench_default_final2_stdout.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['phases_ms'], d['two_level']['deflation_ms'], d['two_level'].get('deflation_mfma_mu8',{}).get('ms'), d['configs_1']['applies_per_sec'], d['configs_3_share'].get('value'), d['configs_4_share'].get('value'))"
tail -3 gpurun_out/r03/bench_default_final2_stderr.log
