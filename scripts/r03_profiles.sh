#!/bin/bash
# Round-3 measurement artefacts on the GPU box (one gpurun call), configs[2] = the default workload of bench.py.
# Output gpurun_out/r03/: default bench line, kernel-trace summary of the same workload (+ per-solve spans: the groups of
# subdomains overlap on four streams), FETCH_SIZE / WRITE_SIZE passes (separate runs, as the guide prescribes), per-level tables.
cd "$(dirname "$0")/.." || exit 1
R=$PWD
out=$R/gpurun_out/r03
if [ -n "$SECONDARY_ONLY" ]; then mkdir -p "$out"; else
rm -rf "$out" && mkdir -p "$out"
if [ -z "$SKIP_DEFAULT" ]; then
  ( time timeout 1200 python bench.py ) > "$out/bench_default_stdout.log" 2> "$out/bench_default_stderr.log"
  grep '^{"metric"' "$out/bench_default_stdout.log" | tail -1 > "$out/bench_default_stdout.json"
fi
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 2 --no-cpu-baseline --no-configs-1 --no-geneo"
timeout 900 rocprofv3 --kernel-trace --stats -d $out/trace -o t -- python $R/bench.py $ARGS > $out/trace.log 2>&1
db=$(find $out/trace -name "*.db" | head -1)
python $R/scripts/prof_summary.py "$db" rocprofv3 --kernel-trace --stats -- python bench.py $ARGS > $out/kernel_stats.csv
python $R/scripts/prof_sweeps.py "$db" 4 > $out/sptrsv_sweeps.csv
grep '^{"metric"' $out/trace.log | tail -1 > $out/trace_bench_line.json
rm -rf $out/trace
PARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-gmres --no-two-level --no-configs-1 --no-shares"
if [ -z "$WITH_PMC" ]; then PMC_CTRS=""; else PMC_CTRS="FETCH_SIZE WRITE_SIZE"; fi # round 3: the passes faulted inside the factorisation under --pmc (gpurun_out/r03_failed_pmc); scripts/r03_pmc_c2.sh collects them at configs[1] on a host-level factorisation
for ctr in $PMC_CTRS; do
  timeout 900 rocprofv3 --pmc $ctr -d $out/pmc_$ctr -o p -- python $R/bench.py $PARGS > $out/pmc_$ctr.log 2>&1
  pdb=$(find $out/pmc_$ctr -name "*.db" | head -1)
  python $R/scripts/pmc_summary.py "$pdb" > $out/pmc_$ctr.csv
  python $R/scripts/pmc_total.py "$pdb" 4 > $out/pmc_${ctr}_last_solve.txt
  grep '^{"metric"' $out/pmc_$ctr.log | tail -1 > $out/pmc_${ctr}_bench_line.json
  rm -rf $out/pmc_$ctr
done
cd $R
timeout 600 python scripts/sweep_plan.py --grid 256 --levels "HPDDM_HIP_STREAMS=1" "" > $out/levels_c3.txt 2>&1
timeout 300 python scripts/sweep_plan.py --grid 128 --levels --reps 30 "HPDDM_HIP_STREAMS=1" "" > $out/levels_c2.txt 2>&1
fi   # SECONDARY_ONLY=1: only the three lines below (the other workloads of bench.py), the rest of gpurun_out/r03 is kept
cd $R
timeout 300 python bench.py --grid 128 --no-two-level --steps 50 > $out/bench_c2_stdout.log 2>&1
grep '^{"metric"' $out/bench_c2_stdout.log | tail -1 > $out/bench_c2_stdout.json
timeout 300 python bench.py --problem helmholtz --grid 64 --mu 8 --steps 20 > $out/bench_c4share_helmholtz_stdout.log 2>&1
grep '^{"metric"' $out/bench_c4share_helmholtz_stdout.log | tail -1 > $out/bench_c4share_helmholtz_stdout.json
timeout 300 python bench.py --problem elasticity --grid 64 --geneo-nu 12 --steps 20 --no-cpu-baseline > $out/bench_c3share_elasticity_stdout.log 2>&1
grep '^{"metric"' $out/bench_c3share_elasticity_stdout.log | tail -1 > $out/bench_c3share_elasticity_stdout.json
timeout 120 python scripts/sweep_plan.py --helmholtz 64,64,128 --mu 1,4,8 --levels --reps 20 "HPDDM_HIP_STREAMS=1" "" > $out/levels_c4share_helmholtz.txt 2>&1
ls -la $out; tail -2 $out/sptrsv_sweeps.csv; cat $out/pmc_*_last_solve.txt | head -4; grep "^==" $out/levels_c*.txt
