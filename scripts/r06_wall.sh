#!/bin/bash
# developer aid (round 6): the one-pass top blocks (W for every wide supernode of the device levels) against the roots only and against none
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06
out=gpurun_out/r06/wall.txt; : > $out
timeout 600 python -m pytest tests/test_gpu_edge_cases.py -q -x -k "root_of_the_tree or collapsed or pivot" 2>&1 | tail -3 | tee -a $out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -3 | tee -a $out
for g in 128 256; do
  for v in "" "HPDDM_HIP_ROOT_W_ONLY=1"; do
    echo "## grid $g  [$v]" | tee -a $out
    env $v timeout 900 python scripts/sweep_plan.py --grid $g --mu 1 --reps 30 --levels "HPDDM_HIP_STREAMS=1" "" "HPDDM_HIP_W_MIN=512" "HPDDM_HIP_W_MIN=1024" "HPDDM_HIP_W_MIN=2048" 2>&1 | grep -E "^==|^setup|total|rror|level 500" | tee -a $out
  done
done
