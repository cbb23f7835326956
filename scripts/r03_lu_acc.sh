#!/bin/bash
mkdir -p gpurun_out/r03
export OMP_NUM_THREADS=8
for N in 33 49 65; do
HPDDM_HIP_VERBOSE=1 timeout 300 python scripts/time_numfact.py $N lu device,host 2>&1 | grep -E "probe|residual" 
done > gpurun_out/r03/lu_acc.log 2>&1
cat gpurun_out/r03/lu_acc.log
