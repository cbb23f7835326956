#!/bin/bash
mkdir -p gpurun_out/r03
export OMP_NUM_THREADS=8
timeout 400 python scripts/time_deflation.py 256 "-hpddm_hip_deflation_pairs 0" "-hpddm_hip_deflation_pairs 1 -hpddm_hip_deflation_blocks_per_cu 3" "-hpddm_hip_deflation_blocks_per_cu 2" "-hpddm_hip_deflation_blocks_per_cu 4" "-hpddm_hip_deflation_blocks_per_cu 6" > gpurun_out/r03/defl_sweep.log 2>&1
cat gpurun_out/r03/defl_sweep.log | tail -14
