#!/bin/bash
# round 4, fourth GPU call: whole GPU suite on the pipelined set-up (two factorisations / eigenproblems in flight), default bench line,
# kernel trace of one 129^3 factorisation
cd "$(dirname "$0")/.." || exit 1
R=$PWD
out=$R/gpurun_out/r04
mkdir -p $out
export OMP_NUM_THREADS=8
timeout 1500 python -m pytest tests -q -m gpu > $out/call4_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -8 $out/call4_gpu_tests.log | cut -c1-300
( time timeout 1200 python bench.py ) > $out/call4_bench_default.log 2> $out/call4_bench_default.err
grep '^{"metric"' $out/call4_bench_default.log | tail -1 > $out/call4_bench_default.json
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04/call4_bench_default.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'setup', d['config']['setup_seconds'], d['config']['setup_seconds_by_phase_summed_over_subdomains'])
print('two_level', {k: d['two_level'][k] for k in ('coarse_setup_seconds', 'coarse_space_seconds', 'deflation_ms')}, d['two_level']['gmres'])
print('host_pointer', d.get('host_pointer_boundary'))
c = d['cpu_baseline']; print('cpu', c['value'], c['cores'], c['sample'][:900])
print('c1', d['configs_1']['roofline']['frac'], d['configs_1']['apply_ms'], d['configs_1']['setup_seconds'])
for k in ('configs_3_share', 'configs_4_share'):
    s = d.get(k, {}); print(k, s.get('value'), s.get('ms_per_step'), (s.get('roofline') or {}).get('frac'), s.get('setup_seconds'), (s.get('two_level') or {}).get('coarse_space_seconds'), s.get('error'))
PY
tail -3 $out/call4_bench_default.err
cd /tmp && export TMPDIR=/tmp
HPDDM_HIP_PROFILE=1 timeout 400 rocprofv3 --kernel-trace --stats -d $out/tr_nf -o t -- python $R/scripts/time_numfact.py 129 chol device > $out/call4_numfact129.log 2>&1
db=$(find $out/tr_nf -name "*.db" | head -1)
python $R/scripts/prof_summary.py "$db" rocprofv3 --kernel-trace --stats -- python scripts/time_numfact.py 129 chol device > $out/call4_numfact129_kernel_stats.csv
rm -rf $out/tr_nf
head -22 $out/call4_numfact129_kernel_stats.csv | cut -c1-220
grep -E "numfact\]|device chol" $out/call4_numfact129.log | tail -12
