#!/bin/bash
# round 5, sixth GPU call: 16-byte loads in the deflation kernels, one-fetch tiles in the 16-column engine
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r05f
rm -rf $out && mkdir -p $out
export OMP_NUM_THREADS=8
timeout 300 python scripts/check_sptrsv.py 9,24 > $out/check_sptrsv.log 2>&1; echo "check rc=$?"; tail -1 $out/check_sptrsv.log
timeout 600 python -m pytest tests/test_sptrsv16.py tests/test_complex.py tests/test_helmholtz.py tests/test_gpu_parity.py -q -m gpu -x > $out/gpu_tests_subset.log 2>&1; echo "gpu tests rc=$?"; tail -6 $out/gpu_tests_subset.log | cut -c1-300
MUS=1,8 timeout 400 python scripts/time_deflation.py 256 "" > $out/deflation_256.txt 2>&1; tail -2 $out/deflation_256.txt
timeout 200 python scripts/sweep_plan.py --helmholtz 64,64,128 --mu 1,8 --levels --reps 20 "HPDDM_HIP_STREAMS=1" "" > $out/levels_c4share_helmholtz.txt 2>&1
timeout 300 python scripts/sweep_plan.py --grid 128 --levels --reps 30 "HPDDM_HIP_STREAMS=1" "" > $out/levels_c2.txt 2>&1
grep "^==" $out/levels_c*.txt
