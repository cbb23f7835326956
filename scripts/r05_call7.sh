#!/bin/bash
# round 5, seventh GPU call: plan knobs at configs[2] (8 x 129^3), one factorisation
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r05g
rm -rf $out && mkdir -p $out
export OMP_NUM_THREADS=8
CF=("" "HPDDM_HIP_BWD_MINROWS=256" "HPDDM_HIP_FWD_ROWS=32" "HPDDM_HIP_FWD_TILE_KB=512" "HPDDM_HIP_FWD_TILE_KB=256" "HPDDM_HIP_LDS=8192" "HPDDM_HIP_LDS=2048" "HPDDM_HIP_BWD_WANT=6144" "HPDDM_HIP_BWD_WANT=1536" "HPDDM_HIP_FWD_WANT=2048" "HPDDM_HIP_STREAMS=2" "HPDDM_HIP_STREAMS=1")
timeout 900 python scripts/sweep_plan.py --grid 256 --reps 10 "${CF[@]}" > $out/knobs_c3.txt 2>&1
grep "^==\|setup" $out/knobs_c3.txt
