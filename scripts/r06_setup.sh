#!/bin/bash
# developer aid (round 6): where the set-up of configs[2] goes -- phases of call_numfact, per-level times of the host and device levels
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06
HPDDM_HIP_PROFILE=1 timeout 600 python scripts/sweep_plan.py --grid 256 --reps 3 --options "-hpddm_hip_numfact_threads 1" "" > gpurun_out/r06/setup_profile_1thread.txt 2>&1
HPDDM_HIP_PROFILE=1 timeout 600 python scripts/sweep_plan.py --grid 256 --reps 3 "" > gpurun_out/r06/setup_profile_2threads.txt 2>&1
grep -E "call_numfact|^setup" gpurun_out/r06/setup_profile_*.txt
