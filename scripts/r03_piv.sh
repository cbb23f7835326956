#!/bin/bash
# pivoting inside the tiles: new tests first, the LU timing, then the whole GPU suite
mkdir -p gpurun_out/r03
export OMP_NUM_THREADS=8
timeout 600 python -m pytest tests/test_pivoting.py -q -m gpu -x > gpurun_out/r03/piv_tests.log 2>&1; echo "pivoting tests rc=$?"; tail -15 gpurun_out/r03/piv_tests.log
timeout 300 python scripts/time_numfact.py 65 lu,ldlt device > gpurun_out/r03/numfact65_piv.log 2>&1; tail -3 gpurun_out/r03/numfact65_piv.log
timeout 1200 python -m pytest tests -q -m gpu -x > gpurun_out/r03/gpu_tests_piv.log 2>&1; echo "gpu tests rc=$?"; tail -5 gpurun_out/r03/gpu_tests_piv.log
