#!/bin/bash
# Collect the round's measurement artefacts on the GPU box (one gpurun call): default bench line, kernel trace summary,
# per-level table, FETCH_SIZE / WRITE_SIZE passes (separate runs, as the guide prescribes).  Output: gpurun_out/final/
cd "$(dirname "$0")/.." || exit 1
R=$PWD
out=$R/gpurun_out/final
rm -rf "$out" && mkdir -p "$out"
timeout 600 python bench.py > "$out/bench_default_stdout.log" 2> "$out/bench_default_stderr.log"
tail -1 "$out/bench_default_stdout.log" | cut -c1-400
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 3 --no-cpu-baseline --no-gmres --no-two-level"
HPDDM_HIP_LEVEL_STATS=$out/levels.txt timeout 300 rocprofv3 --kernel-trace --stats -d $out/trace -o t -- python $R/bench.py $ARGS > $out/trace.log 2>&1
db=$(find $out/trace -name "*.db" | head -1)
n=$(grep '^{"metric"' $out/trace.log | tail -1 | python -c 'import sys, json; print(int(json.loads(sys.stdin.readline())["config"]["launches_per_sptrsv"]) - 2)')
python $R/scripts/prof_summary.py "$db" rocprofv3 --kernel-trace --stats -- python bench.py $ARGS > $out/kernel_stats.csv
python $R/scripts/prof_levels.py "$db" "$n" $out/levels.txt > $out/sptrsv_levels.txt
grep '^{"metric"' $out/trace.log | tail -1 > $out/trace_bench_line.json
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $ctr -d $out/pmc_$ctr -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-gmres --no-two-level > $out/pmc_$ctr.log 2>&1
  pdb=$(find $out/pmc_$ctr -name "*.db" | head -1)
  python $R/scripts/pmc_summary.py "$pdb" > $out/pmc_$ctr.csv
  python $R/scripts/pmc_levels.py "$pdb" "$n" $out/levels.txt > $out/pmc_${ctr}_levels.csv
done
rm -rf $out/trace $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
ls -la $out
tail -3 $out/sptrsv_levels.txt; tail -2 $out/pmc_FETCH_SIZE_levels.csv; tail -2 $out/pmc_WRITE_SIZE_levels.csv
