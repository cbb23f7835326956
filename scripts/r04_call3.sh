#!/bin/bash
# round 4, third GPU call: the tests added since call 2 (inertia / estimate_nu, complex RHS deflation, complex custom operator), then the
# default bench line (configs[2]) with the sampled CPU baseline and the host-pointer boundary key
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r04
mkdir -p $out
export OMP_NUM_THREADS=8
timeout 900 python -m pytest tests/test_complex.py tests/test_pivoting.py tests/test_helmholtz.py "tests/test_gpu_parity.py::test_bgmres_matches_reference" tests/test_gpu_parity.py::test_custom_operator_callbacks tests/test_gpu_dropin.py -q -m gpu > $out/call3_tests.log 2>&1; echo "tests rc=$?"; tail -12 $out/call3_tests.log | cut -c1-300
( time timeout 1200 python bench.py ) > $out/call3_bench_default.log 2> $out/call3_bench_default.err
grep '^{"metric"' $out/call3_bench_default.log | tail -1 > $out/call3_bench_default.json
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04/call3_bench_default.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'setup', d['config']['setup_seconds'], d['config']['setup_seconds_by_phase_summed_over_subdomains'])
print('phases', d['phases_ms'])
print('two_level', {k: v for k, v in d['two_level'].items() if k not in ('kernel',)})
print('host_pointer', d.get('host_pointer_boundary'))
c = d['cpu_baseline']; print('cpu', c['value'], c['cores'], c['sample'][:600])
print('c1', d['configs_1']['roofline']['frac'], d['configs_1']['apply_ms'])
for k in ('configs_3_share', 'configs_4_share'):
    s = d.get(k, {}); print(k, s.get('value'), s.get('ms_per_step'), (s.get('roofline') or {}).get('frac'), (s.get('two_level') or {}).get('gmres'), s.get('error'))
PY
tail -5 $out/call3_bench_default.err
