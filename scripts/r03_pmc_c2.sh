#!/bin/bash
# HBM traffic of one batched SpTRSV at configs[1] (128^3, 8 subdomains): FETCH_SIZE / WRITE_SIZE in separate rocprofv3 --pmc runs
# (as scripts/r03_profiles.sh does for configs[2]); scripts/r03_collect.py turns gpurun_out/r03c2/ into profiles/r03_pmc_traffic_c2.json
cd "$(dirname "$0")/.." || exit 1
R=$PWD
out=$R/gpurun_out/r03c2
rm -rf "$out" && mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
# (the factorisation on the host levels only: under rocprofv3 --pmc the pinned upload ring of the device levels faulted inside
# hipMemcpyAsync in round 3, gpurun_out/r03_failed_pmc; the sweeps measured are the same)
export HPDDM_HIP_HOST_FACTOR=1
PARGS="--grid 128 --steps 3 --warmup 1 --no-cpu-baseline --no-gmres --no-two-level --no-shares"
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 240 rocprofv3 --pmc $ctr -d $out/pmc_$ctr -o p -- python $R/bench.py $PARGS > $out/pmc_$ctr.log 2>&1
  pdb=$(find $out/pmc_$ctr -name "*.db" | head -1)
  python $R/scripts/pmc_summary.py "$pdb" > $out/pmc_$ctr.csv
  python $R/scripts/pmc_total.py "$pdb" 4 > $out/pmc_${ctr}_last_solve.txt
  grep '^{"metric"' $out/pmc_$ctr.log | tail -1 > $out/pmc_${ctr}_bench_line.json
  rm -rf $out/pmc_$ctr
done
cat $out/pmc_*_last_solve.txt | head -8
