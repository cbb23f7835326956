#!/bin/bash
# round 5, tenth GPU call: condensed leaves two to a wavefront, forward / backward / both / none on one box
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r05j
rm -rf $out && mkdir -p $out
export OMP_NUM_THREADS=8
timeout 600 python scripts/sweep_plan.py --grid 128 --reps 30 --levels "HPDDM_HIP_STREAMS=1" "HPDDM_HIP_STREAMS=1 HPDDM_HIP_LEAF_PAIRS=0" "HPDDM_HIP_STREAMS=1 HPDDM_HIP_LEAF_PAIRS=1" "HPDDM_HIP_STREAMS=1 HPDDM_HIP_LEAF_PAIRS=2" "" "HPDDM_HIP_LEAF_PAIRS=0" "HPDDM_HIP_LEAF_PAIRS=1" "HPDDM_HIP_LEAF_PAIRS=2" "" "HPDDM_HIP_LEAF_PAIRS=0" > $out/levels_c2.txt 2>&1
grep "^==\|level  0" $out/levels_c2.txt
timeout 600 python scripts/sweep_plan.py --helmholtz 64,64,128 --mu 1,8 --reps 30 --levels "HPDDM_HIP_STREAMS=1" "HPDDM_HIP_STREAMS=1 HPDDM_HIP_LEAF_PAIRS=0" "" "HPDDM_HIP_LEAF_PAIRS=0" "HPDDM_HIP_LEAF_PAIRS=1" "HPDDM_HIP_LEAF_PAIRS=2" "" "HPDDM_HIP_LEAF_PAIRS=0" > $out/levels_h1.txt 2>&1
grep "^==\|level  0" $out/levels_h1.txt
