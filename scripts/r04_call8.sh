#!/bin/bash
# after scripts/r04_final.sh: GenEO helper kernels reworked (tests that use them), per-level table of the Helmholtz share, and the default
# bench line once more (it now quotes this round's PMC traffic file)
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r04f
mkdir -p $out
export OMP_NUM_THREADS=8
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_elasticity.py tests/test_pivoting.py tests/test_gpu_full_size.py -q -m gpu > $out/call8_tests.log 2>&1; echo "tests rc=$?"; tail -4 $out/call8_tests.log | cut -c1-300
timeout 200 python scripts/sweep_plan.py --helmholtz 64,64,128 --mu 1,8 --levels --reps 20 "HPDDM_HIP_STREAMS=1" "" > $out/levels_c4share_helmholtz.txt 2>&1
grep "^==" $out/levels_c4share_helmholtz.txt
( time timeout 1200 python bench.py ) > "$out/bench_default_stdout.log" 2> "$out/bench_default_stderr.log"
grep '^{"metric"' "$out/bench_default_stdout.log" | tail -1 > "$out/bench_default_stdout.json"
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04f/bench_default_stdout.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], d['roofline'].get('traffic'), d['roofline'].get('traffic_source'))
print('setup', d['config']['setup_seconds'], 'geneo', d['two_level']['coarse_space_seconds'], 'coarse', d['two_level']['coarse_setup_seconds'], 'gmres', d['two_level']['gmres']['iterations'], d['one_level']['gmres']['iterations'])
print('c1', d['configs_1']['roofline']['frac'], d['configs_1']['roofline'].get('traffic_source'))
PY
tail -3 $out/bench_default_stderr.log
