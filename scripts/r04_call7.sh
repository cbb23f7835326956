#!/bin/bash
# after the host-side set-up work (matrix expansion side by side, coarse assembly block rows in parallel): the tests it touches, set-up times
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r04
mkdir -p $out
export OMP_NUM_THREADS=8
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_distributed.py tests/test_elasticity.py tests/test_complex.py tests/test_helmholtz.py tests/test_gpu_edge_cases.py tests/test_pivoting.py -q -m gpu > $out/call7_tests.log 2>&1; echo "tests rc=$?"; tail -5 $out/call7_tests.log | cut -c1-300
timeout 600 python bench.py --no-cpu-baseline --no-configs-1 --no-shares --steps 10 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('setup', d['config']['setup_seconds'], d['config']['setup_seconds_by_phase_summed_over_subdomains'], 'geneo', d['two_level']['coarse_space_seconds'], 'coarse', d['two_level']['coarse_setup_seconds'], 'apply', d['ms_per_step'], 'gmres', d['two_level']['gmres']['iterations'], d['one_level']['gmres']['iterations'])"
