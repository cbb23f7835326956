#!/bin/bash
mkdir -p gpurun_out/r03
export OMP_NUM_THREADS=8
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_complex.py tests/test_gpu_dropin.py -q -m gpu -x -k "deflat or two_level or coarse or complex or geneo or panel or hook" > gpurun_out/r03/mfma_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r03/mfma_tests.log
MUS=1,4,8,16 timeout 300 python scripts/time_deflation.py 256 > gpurun_out/r03/defl_mfma_times2.log 2>&1; tail -5 gpurun_out/r03/defl_mfma_times2.log
timeout 300 python scripts/time_deflation.py helmholtz > gpurun_out/r03/helm_phases2.log 2>&1; tail -2 gpurun_out/r03/helm_phases2.log
