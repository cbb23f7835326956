#!/bin/bash
# Round-4 measurement artefacts on the GPU box (one gpurun call).  Output gpurun_out/r04f/: the whole GPU suite, the default bench
# line (configs[2]), the kernel-trace summary of the same workload (+ per-solve spans: the groups of subdomains overlap on four
# streams), the other workloads of bench.py stand-alone, per-level tables, and LAST the FETCH_SIZE / WRITE_SIZE passes (separate
# runs, as the guide prescribes; every pass under its own timeout: a pass that hangs costs its timeout, nothing else).
cd "$(dirname "$0")/.." || exit 1
R=$PWD
out=$R/gpurun_out/r04f
rm -rf "$out" && mkdir -p "$out"
export OMP_NUM_THREADS=8
timeout 1500 python -m pytest tests -q -m gpu > $out/gpu_tests_final.log 2>&1; echo "gpu tests rc=$?"; tail -4 $out/gpu_tests_final.log | cut -c1-300
( time timeout 1200 python bench.py ) > "$out/bench_default_stdout.log" 2> "$out/bench_default_stderr.log"
grep '^{"metric"' "$out/bench_default_stdout.log" | tail -1 > "$out/bench_default_stdout.json"
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-configs-1 --no-shares"
timeout 900 rocprofv3 --kernel-trace --stats -d $out/trace -o t -- python $R/bench.py $ARGS > $out/trace.log 2>&1
db=$(find $out/trace -name "*.db" | head -1)
python $R/scripts/prof_summary.py "$db" rocprofv3 --kernel-trace --stats -- python bench.py $ARGS > $out/kernel_stats.csv
python $R/scripts/prof_sweeps.py "$db" 4 > $out/sptrsv_sweeps.csv
grep '^{"metric"' $out/trace.log | tail -1 > $out/trace_bench_line.json
rm -rf $out/trace
cd $R
timeout 300 python bench.py --grid 128 --no-two-level --steps 50 > $out/bench_c2_stdout.log 2>&1
grep '^{"metric"' $out/bench_c2_stdout.log | tail -1 > $out/bench_c2_stdout.json
timeout 300 python bench.py --problem helmholtz --grid 64 --mu 8 --geneo-nu 12 --steps 20 > $out/bench_c4share_helmholtz_stdout.log 2>&1
grep '^{"metric"' $out/bench_c4share_helmholtz_stdout.log | tail -1 > $out/bench_c4share_helmholtz_stdout.json
timeout 300 python bench.py --problem elasticity --grid 64 --geneo-nu 12 --steps 20 --no-cpu-baseline > $out/bench_c3share_elasticity_stdout.log 2>&1
grep '^{"metric"' $out/bench_c3share_elasticity_stdout.log | tail -1 > $out/bench_c3share_elasticity_stdout.json
BENCH_SHARE_GPU=1 timeout 400 python bench.py --gpus 4 --problem helmholtz --grid 32 --mu 8 --geneo-nu 6 --no-cpu-baseline > $out/share4_helmholtz.log 2>&1
grep '^{"metric"' $out/share4_helmholtz.log | tail -1 > $out/share4_helmholtz.json
BENCH_SHARE_GPU=1 timeout 400 python bench.py --gpus 8 --problem elasticity --grid 16 --geneo-nu 6 --no-cpu-baseline > $out/share8_elasticity.log 2>&1
grep '^{"metric"' $out/share8_elasticity.log | tail -1 > $out/share8_elasticity.json
timeout 300 python scripts/sweep_plan.py --grid 128 --levels --reps 30 "HPDDM_HIP_STREAMS=1" "" > $out/levels_c2.txt 2>&1
timeout 120 python scripts/sweep_plan.py --helmholtz 64,64,128 --mu 1,8 --levels --reps 20 "HPDDM_HIP_STREAMS=1" "" > $out/levels_c4share_helmholtz.txt 2>&1
timeout 600 python scripts/sweep_plan.py --grid 256 --levels "HPDDM_HIP_STREAMS=1" "" > $out/levels_c3.txt 2>&1
ls -la $out | head -40; tail -2 $out/sptrsv_sweeps.csv; grep "^==" $out/levels_c*.txt
# ---- PMC passes, last (HPDDM_HIP_UPLOAD_UNPINNED: under --pmc the copy from the pinned upload ring faulted in round 3) ----
cd /tmp
export HPDDM_HIP_UPLOAD_UNPINNED=1
PARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-gmres --no-two-level --no-configs-1 --no-shares --options=-hpddm_hip_numfact_threads=1"
mkdir -p $out/c2
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 240 rocprofv3 --pmc $ctr -d $out/c2/pmc_$ctr -o p -- python $R/bench.py --grid 128 $PARGS > $out/c2/pmc_$ctr.log 2>&1 || { echo "c2 $ctr pass failed"; tail -3 $out/c2/pmc_$ctr.log; rm -rf $out/c2/pmc_$ctr; continue; }
  pdb=$(find $out/c2/pmc_$ctr -name "*.db" | head -1)
  python $R/scripts/pmc_summary.py "$pdb" > $out/c2/pmc_$ctr.csv
  python $R/scripts/pmc_total.py "$pdb" 4 > $out/c2/pmc_${ctr}_last_solve.txt
  grep '^{"metric"' $out/c2/pmc_$ctr.log | tail -1 > $out/c2/pmc_${ctr}_bench_line.json
  rm -rf $out/c2/pmc_$ctr
done
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $ctr -d $out/pmc_$ctr -o p -- python $R/bench.py $PARGS > $out/pmc_$ctr.log 2>&1 || { echo "c3 $ctr pass failed"; tail -3 $out/pmc_$ctr.log; rm -rf $out/pmc_$ctr; continue; }
  pdb=$(find $out/pmc_$ctr -name "*.db" | head -1)
  python $R/scripts/pmc_summary.py "$pdb" > $out/pmc_$ctr.csv
  python $R/scripts/pmc_total.py "$pdb" 4 > $out/pmc_${ctr}_last_solve.txt
  grep '^{"metric"' $out/pmc_$ctr.log | tail -1 > $out/pmc_${ctr}_bench_line.json
  rm -rf $out/pmc_$ctr
done
cat $out/c2/pmc_*_last_solve.txt $out/pmc_*_last_solve.txt 2>/dev/null | grep -v "^#" | head -8
