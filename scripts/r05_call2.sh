#!/bin/bash
# round 5, second GPU call: early tile with everything in flight, per-tile descriptors, compact hand-over in the 16-column engine
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r05b
rm -rf $out && mkdir -p $out
export OMP_NUM_THREADS=8
timeout 300 python scripts/check_sptrsv.py 9,24 > $out/check_sptrsv.log 2>&1; echo "check rc=$?"; grep -c "^ok" $out/check_sptrsv.log; grep "FAIL\|Error\|error" $out/check_sptrsv.log | head -20; tail -1 $out/check_sptrsv.log
export HPDDM_HIP_KEEP_FT=1
timeout 300 python scripts/sweep_plan.py --grid 128 --levels --reps 30 "HPDDM_HIP_STREAMS=1" "" "HPDDM_HIP_LEAF_TILES=0" > $out/levels_c2.txt 2>&1
timeout 200 python scripts/sweep_plan.py --helmholtz 64,64,128 --mu 1,8 --levels --reps 20 "HPDDM_HIP_STREAMS=1" "" > $out/levels_c4share_helmholtz.txt 2>&1
unset HPDDM_HIP_KEEP_FT
timeout 600 python scripts/sweep_plan.py --grid 256 --levels "HPDDM_HIP_STREAMS=1" "" > $out/levels_c3.txt 2>&1
grep "^==" $out/levels_c*.txt
timeout 1200 python -m pytest tests -q -m gpu > $out/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -25 $out/gpu_tests.log | cut -c1-300
