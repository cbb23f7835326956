#!/bin/bash
mkdir -p gpurun_out/r03
export OMP_NUM_THREADS=8
HPDDM_HIP_PROFILE=1 timeout 300 python scripts/time_numfact.py 129 chol device > gpurun_out/r03/numfact129_v2b.log 2>&1; grep -E "device levels|numfact " gpurun_out/r03/numfact129_v2b.log | tail -6
