#!/bin/bash
# experiments on the set-up: panel width of the device levels, factorisations in flight
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r04
mkdir -p $out
export OMP_NUM_THREADS=8
for pw in 256 384 512 768; do
  echo "== HPDDM_HIP_PANEL=$pw"
  HPDDM_HIP_PANEL=$pw timeout 300 python scripts/time_numfact.py 129 chol,ldlt device 2>&1 | grep -E "^device"
done 2>&1 | tee $out/call6_panel_width.txt
for nt in 1 2 3; do
  echo "== bench set-up, -hpddm_hip_numfact_threads $nt"
  timeout 600 python bench.py --no-cpu-baseline --no-configs-1 --no-shares --no-gmres --steps 5 --options=-hpddm_hip_numfact_threads=$nt 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('setup', d['config']['setup_seconds'], d['config']['setup_seconds_by_phase_summed_over_subdomains'], 'geneo', d['two_level']['coarse_space_seconds'], 'coarse', d['two_level']['coarse_setup_seconds'], 'apply', d['ms_per_step'])"
done 2>&1 | tee $out/call6_numfact_threads.txt
