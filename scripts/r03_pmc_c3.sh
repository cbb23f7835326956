#!/bin/bash
# HBM traffic of one batched SpTRSV at configs[2] (the default workload): FETCH_SIZE / WRITE_SIZE in separate rocprofv3 --pmc runs.
# HPDDM_HIP_UPLOAD_UNPINNED: the hand-over lists of the device levels by plain hipMemcpy (under --pmc the copy from the pinned ring
# faulted inside hipMemcpyAsync, gpurun_out/r03_failed_pmc); a pass that does not finish in 4 minutes is given up.
cd "$(dirname "$0")/.." || exit 1
R=$PWD
out=$R/gpurun_out/r03
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
export HPDDM_HIP_UPLOAD_UNPINNED=1
PARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-gmres --no-two-level --no-configs-1 --no-shares"
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 240 rocprofv3 --pmc $ctr -d $out/pmc_$ctr -o p -- python $R/bench.py $PARGS > $out/pmc_$ctr.log 2>&1 || { echo "$ctr pass failed"; tail -3 $out/pmc_$ctr.log; rm -rf $out/pmc_$ctr; exit 0; }
  pdb=$(find $out/pmc_$ctr -name "*.db" | head -1)
  python $R/scripts/pmc_summary.py "$pdb" > $out/pmc_$ctr.csv
  python $R/scripts/pmc_total.py "$pdb" 4 > $out/pmc_${ctr}_last_solve.txt
  grep '^{"metric"' $out/pmc_$ctr.log | tail -1 > $out/pmc_${ctr}_bench_line.json
  rm -rf $out/pmc_$ctr
done
cat $out/pmc_*_last_solve.txt | head -8
