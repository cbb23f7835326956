#!/bin/bash
mkdir -p gpurun_out/r03
export OMP_NUM_THREADS=8
timeout 900 python -m pytest tests/test_gpu_dropin.py -m gpu -x -q > gpurun_out/r03/dropin_tests.log 2>&1; echo "dropin rc=$?"; grep -E "passed|failed|Error|assert" gpurun_out/r03/dropin_tests.log | head; tail -30 gpurun_out/r03/dropin_tests.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr" | tail -15
