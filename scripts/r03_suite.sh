#!/bin/bash
mkdir -p gpurun_out/r03
export OMP_NUM_THREADS=8
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r03/gpu_tests_engine.log 2>&1; echo "gpu tests rc=$?"; grep -E "passed|failed" gpurun_out/r03/gpu_tests_engine.log
