#!/bin/bash
# round 5, third GPU call: tile-size knobs of the wide levels (a level ends with the tail of its last tiles) at configs[1] and on the
# Helmholtz share, one factorisation each
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r05c
rm -rf $out && mkdir -p $out
export OMP_NUM_THREADS=8
CF=("" "HPDDM_HIP_FWD_ROWS=32" "HPDDM_HIP_FWD_ROWS=16" "HPDDM_HIP_FWD_TILE_KB=128" "HPDDM_HIP_FWD_TILE_KB=64" "HPDDM_HIP_FWD_WANT=2048" "HPDDM_HIP_FWD_WANT=8192" \
    "HPDDM_HIP_BWD_MINROWS=128" "HPDDM_HIP_BWD_MINROWS=64" "HPDDM_HIP_BWD_WANT=6144" "HPDDM_HIP_BWD_WANT=12288 HPDDM_HIP_BWD_MAXPARTS=64" "HPDDM_HIP_BWD_MINROWS=128 HPDDM_HIP_BWD_WANT=6144 HPDDM_HIP_BWD_MAXPARTS=64" \
    "HPDDM_HIP_FWD_ROWS=16 HPDDM_HIP_BWD_MINROWS=128 HPDDM_HIP_BWD_WANT=6144 HPDDM_HIP_BWD_MAXPARTS=64" "HPDDM_HIP_FWD_TILE_KB=128 HPDDM_HIP_BWD_MINROWS=128 HPDDM_HIP_BWD_WANT=6144 HPDDM_HIP_BWD_MAXPARTS=64" \
    "HPDDM_HIP_STREAMS=1" "HPDDM_HIP_STREAMS=1 HPDDM_HIP_FWD_ROWS=16 HPDDM_HIP_BWD_MINROWS=128 HPDDM_HIP_BWD_WANT=6144 HPDDM_HIP_BWD_MAXPARTS=64")
timeout 400 python scripts/sweep_plan.py --grid 128 --levels --reps 30 "${CF[@]}" > $out/knobs_c2.txt 2>&1
timeout 300 python scripts/sweep_plan.py --helmholtz 64,64,128 --mu 1,8 --levels --reps 20 "${CF[@]}" > $out/knobs_c4share_helmholtz.txt 2>&1
grep "^==" $out/knobs_c*.txt
