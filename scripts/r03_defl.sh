#!/bin/bash
mkdir -p gpurun_out/r03
export OMP_NUM_THREADS=8
MUS=4,8,16 timeout 300 python scripts/time_deflation.py 256 "-hpddm_hip_deflation_zt_direct 0" "-hpddm_hip_deflation_zt_direct 1" > gpurun_out/r03/defl_zt2_times.log 2>&1; tail -7 gpurun_out/r03/defl_zt2_times.log
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py tests/test_gpu_edge_cases.py -q -m gpu -x -k "deflat or two_level or coarse or geneo or panel or hook or bgmres or many_right" > gpurun_out/r03/zt2_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r03/zt2_tests.log
