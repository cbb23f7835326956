#!/bin/bash
mkdir -p gpurun_out/r03
export OMP_NUM_THREADS=8
timeout 300 python -m pytest tests/test_complex.py -q -m gpu -x > gpurun_out/r03/gmvz_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r03/gmvz_tests.log
timeout 300 python scripts/time_deflation.py helmholtz > gpurun_out/r03/helm_phases.log 2>&1; tail -4 gpurun_out/r03/helm_phases.log
