#!/bin/bash
# kernel-trace summary + per-level table of the SpTRSV at configs[2] size (256^3): one gpurun call, output gpurun_out/c3/
cd "$(dirname "$0")/.." || exit 1
R=$PWD; out=$R/gpurun_out/c3
rm -rf "$out" && mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
ARGS="--grid 256 --steps 5 --warmup 1 --no-cpu-baseline --no-gmres --no-two-level"
HPDDM_HIP_LEVEL_STATS=$out/levels.txt timeout 900 rocprofv3 --kernel-trace --stats -d $out/trace -o t -- python $R/bench.py $ARGS > $out/trace.log 2>&1
db=$(find $out/trace -name "*.db" | head -1)
n=$(grep '^{"metric"' $out/trace.log | tail -1 | python -c 'import sys, json; print(int(json.loads(sys.stdin.readline())["config"]["launches_per_sptrsv"]) - 2)')
python $R/scripts/prof_summary.py "$db" rocprofv3 --kernel-trace --stats -- python bench.py $ARGS > $out/kernel_stats.csv
python $R/scripts/prof_levels.py "$db" "$n" $out/levels.txt > $out/sptrsv_levels.txt
grep '^{"metric"' $out/trace.log | tail -1 > $out/bench_line.json
rm -rf $out/trace
tail -4 $out/sptrsv_levels.txt
