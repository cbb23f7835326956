#!/bin/bash
# The two HBM-traffic passes only (separate rocprofv3 --pmc runs, as the guide prescribes): FETCH_SIZE and WRITE_SIZE of the last
# batched SpTRSV of `bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-gmres --no-two-level`.  Output: gpurun_out/pmc/
cd "$(dirname "$0")/.." || exit 1
R=$PWD
out=$R/gpurun_out/pmc
rm -rf "$out" && mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-gmres --no-two-level"
for ctr in FETCH_SIZE WRITE_SIZE; do
  HPDDM_HIP_LEVEL_STATS=$out/levels.txt timeout ${PMC_TIMEOUT:-45} rocprofv3 --pmc $ctr -d $out/pmc_$ctr -o p -- python $R/bench.py $ARGS > $out/pmc_$ctr.log 2>&1
  pdb=$(find $out/pmc_$ctr -name "*.db" | head -1)
  n=$(grep '^{"metric"' $out/pmc_$ctr.log | tail -1 | python -c 'import sys, json; print(int(json.loads(sys.stdin.readline())["config"]["launches_per_sptrsv"]) - 2)')
  python $R/scripts/pmc_levels.py "$pdb" "$n" $out/levels.txt > $out/pmc_${ctr}_levels.csv
  grep '^{"metric"' $out/pmc_$ctr.log | tail -1 > $out/pmc_${ctr}_bench_line.json
  rm -rf $out/pmc_$ctr
done
tail -1 $out/pmc_FETCH_SIZE_levels.csv; tail -1 $out/pmc_WRITE_SIZE_levels.csv
