#!/bin/bash
# developer aid (round 6): set-up of configs[2] with more factorisations / eigenproblems in flight on the host side (one device slot)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06
out=gpurun_out/r06/threads_in_flight.txt; : > $out
for v in "X=1" "HPDDM_HIP_GEVP_THREADS=3 HPDDM_HIP_NUMFACT_THREADS=3" "HPDDM_HIP_GEVP_THREADS=4 HPDDM_HIP_NUMFACT_THREADS=4" "HPDDM_HIP_GEVP_THREADS=3"; do
  echo "## $v" | tee -a $out
  env $v timeout 900 python bench.py --no-cpu-baseline --no-configs-1 --no-shares --steps 5 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']
print('setup', c['setup_seconds'], 'coarse_space', d['two_level']['coarse_space_seconds'], 'coarse_setup', d['two_level']['coarse_setup_seconds'], 'frac', round(d['roofline']['frac'],4))" | tee -a $out
done
