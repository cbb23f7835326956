#!/bin/bash
# round 4, fifth GPU call: the factorisation with two fills instead of two per front and no per-front synchronisation of L D L^T;
# which fronts are worth the device (HPDDM_HIP_DEVICE_MIN_H) now that the host levels of the next subdomain hide under the device levels
cd "$(dirname "$0")/.." || exit 1
R=$PWD
out=$R/gpurun_out/r04
mkdir -p $out
export OMP_NUM_THREADS=8
timeout 900 python -m pytest tests/test_pivoting.py tests/test_complex.py tests/test_helmholtz.py tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_elasticity.py -q -m gpu > $out/call5_tests.log 2>&1; echo "tests rc=$?"; tail -6 $out/call5_tests.log | cut -c1-300
for mh in 768 1400 2400 4000; do
  echo "== HPDDM_HIP_DEVICE_MIN_H=$mh"
  HPDDM_HIP_DEVICE_MIN_H=$mh timeout 300 python scripts/time_numfact.py 129 chol,ldlt device 2>&1 | grep -E "^device"
done 2>&1 | tee $out/call5_numfact_min_h.txt
for mh in 768 2400; do
  echo "== bench set-up, HPDDM_HIP_DEVICE_MIN_H=$mh"
  HPDDM_HIP_DEVICE_MIN_H=$mh timeout 600 python bench.py --no-cpu-baseline --no-configs-1 --no-shares --no-gmres --steps 5 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('setup', d['config']['setup_seconds'], d['config']['setup_seconds_by_phase_summed_over_subdomains'], 'geneo', d['two_level']['coarse_space_seconds'], 'apply', d['ms_per_step'])"
done 2>&1 | tee -a $out/call5_numfact_min_h.txt
