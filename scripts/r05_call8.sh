#!/bin/bash
# round 5, eighth GPU call: condensed leaves two to a wavefront -- correctness against SuperLU, then level times with and without
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r05h
rm -rf $out && mkdir -p $out
export OMP_NUM_THREADS=8
timeout 600 python scripts/check_sptrsv.py > $out/check.txt 2>&1; echo "check exit $?"; tail -3 $out/check.txt
timeout 600 python scripts/sweep_plan.py --grid 128 --reps 20 --levels "" "HPDDM_HIP_LEAF_PAIRS=0" > $out/levels_c2.txt 2>&1
grep "^==\|level  0" $out/levels_c2.txt
timeout 600 python scripts/sweep_plan.py --helmholtz 64,64,128 --mu 1 --reps 20 --levels "" "HPDDM_HIP_LEAF_PAIRS=0" > $out/levels_h1.txt 2>&1
grep "^==\|level  0" $out/levels_h1.txt
