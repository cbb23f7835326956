#!/bin/bash
# Round-6 measurement artefacts on the GPU box, one gpurun call:   gpurun --timeout 2400 -- 'bash scripts/r06_profiles.sh [stage ...]'
# Stages (default: all, in this order; every step under its own timeout -- a step that hangs costs its timeout, nothing else):
#   tests       the whole `pytest -m gpu` suite
#   bench       the default bench line (configs[2]) exactly as the driver runs it
#   trace       rocprofv3 --kernel-trace --stats of the same workload + per-solve spans of the overlapping stream groups
#   standalone  the other workloads of bench.py on their own (configs[1], the elasticity and Helmholtz shares, the shared-GPU layouts)
#   levels      per-level tables of the sweeps (scripts/sweep_plan.py), deflation panel times (scripts/time_deflation.py)
#   sq16        SQ counters of the 16-column engine on the Helmholtz share (two --pmc passes around scripts/sweep_plan.py, one group on one stream)
#   pmc         LAST: separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes at configs[1] and configs[2] (MI355X_MICROARCH.md, HBM section)
# Output: gpurun_out/r06/ ; scripts/r06_collect.py copies it into profiles/ under the round's names and derives the traffic files.
cd "$(dirname "$0")/.." || exit 1
R=$PWD
out=$R/gpurun_out/r06
mkdir -p "$out"
export OMP_NUM_THREADS=8
stages=("$@"); [ ${#stages[@]} -eq 0 ] && stages=(tests bench trace standalone levels sq16 pmc)
line() { grep '^{"metric"' "$1" | tail -1 > "$2"; }

for st in "${stages[@]}"; do
  cd "$R"
  case $st in
  tests)
    timeout 1500 python -m pytest tests -q -m gpu > $out/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -4 $out/gpu_tests.log | cut -c1-300 ;;
  bench)
    ( time timeout 1200 python bench.py ) > $out/bench_default_stdout.log 2> $out/bench_default_stderr.log
    line $out/bench_default_stdout.log $out/bench_default_stdout.json; cut -c1-400 $out/bench_default_stdout.json; tail -4 $out/bench_default_stderr.log ;;
  trace)
    cd /tmp && export TMPDIR=/tmp
    ARGS="--no-cpu-baseline --no-configs-1 --no-shares"
    timeout 900 rocprofv3 --kernel-trace --stats -d $out/trace -o t -- python $R/bench.py $ARGS > $out/trace.log 2>&1
    db=$(find $out/trace -name "*.db" | head -1)
    python $R/scripts/prof_summary.py "$db" rocprofv3 --kernel-trace --stats -- python bench.py $ARGS > $out/kernel_stats.csv
    python $R/scripts/prof_sweeps.py "$db" 4 > $out/sptrsv_sweeps.csv
    line $out/trace.log $out/trace_bench_line.json
    rm -rf $out/trace; tail -2 $out/sptrsv_sweeps.csv ;;
  standalone)
    timeout 300 python bench.py --grid 128 --no-two-level --steps 50 > $out/bench_c2_stdout.log 2>&1; line $out/bench_c2_stdout.log $out/bench_c2_stdout.json
    timeout 300 python bench.py --problem helmholtz --grid 64 --mu 8 --geneo-nu 12 --steps 20 > $out/bench_c4share_helmholtz_stdout.log 2>&1; line $out/bench_c4share_helmholtz_stdout.log $out/bench_c4share_helmholtz_stdout.json
    timeout 300 python bench.py --problem elasticity --grid 64 --geneo-nu 12 --steps 20 --no-cpu-baseline > $out/bench_c3share_elasticity_stdout.log 2>&1; line $out/bench_c3share_elasticity_stdout.log $out/bench_c3share_elasticity_stdout.json
    BENCH_SHARE_GPU=1 timeout 400 python bench.py --gpus 4 --problem helmholtz --grid 32 --mu 8 --geneo-nu 6 --no-cpu-baseline > $out/share4_helmholtz.log 2>&1; line $out/share4_helmholtz.log $out/share4_helmholtz.json
    BENCH_SHARE_GPU=1 timeout 400 python bench.py --gpus 8 --problem elasticity --grid 16 --geneo-nu 6 --no-cpu-baseline > $out/share8_elasticity.log 2>&1; line $out/share8_elasticity.log $out/share8_elasticity.json
    for f in $out/bench_c2_stdout.json $out/bench_c4share_helmholtz_stdout.json $out/bench_c3share_elasticity_stdout.json; do cut -c1-200 $f; done ;;
  levels)
    timeout 300 python scripts/sweep_plan.py --grid 128 --levels --reps 30 "HPDDM_HIP_STREAMS=1" "" "HPDDM_HIP_LEAF_TILES=0" "HPDDM_HIP_LEAF_PAIRS=0" > $out/levels_c2.txt 2>&1
    timeout 200 python scripts/sweep_plan.py --helmholtz 64,64,128 --mu 1,8 --levels --reps 20 "HPDDM_HIP_STREAMS=1" "" > $out/levels_c4share_helmholtz.txt 2>&1
    timeout 900 python scripts/sweep_plan.py --grid 256 --levels "HPDDM_HIP_STREAMS=1" "" > $out/levels_c3.txt 2>&1
    MUS=1,8 timeout 300 python scripts/time_deflation.py 256 > $out/deflation_256.txt 2>&1
    grep "^==" $out/levels_c*.txt; tail -4 $out/deflation_256.txt ;;
  sq16)
    cd /tmp && export TMPDIR=/tmp
    for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_MFMA"; do
      tag=$(echo $pass | cut -d" " -f2)
      HPDDM_HIP_STREAMS=1 timeout 300 rocprofv3 --pmc $pass -d $out/sq16_$tag -o p -- python $R/scripts/sweep_plan.py --helmholtz 64,64,128 --mu 8 --reps 40 "HPDDM_HIP_STREAMS=1" > $out/sq16_$tag.log 2>&1
      pdb=$(find $out/sq16_$tag -name "*.db" | head -1)
      python $R/scripts/pmc_summary.py "$pdb" | grep -E "kernel,|sptrsv16" > $out/sq16_$tag.csv
      rm -rf $out/sq16_$tag
    done
    python $R/scripts/sq16_summary.py $out/sq16_SQ_WAVE_CYCLES.csv $out/sq16_SQ_ACTIVE_INST_ANY.csv > $out/engine16_sq_counters.txt; cat $out/engine16_sq_counters.txt | head -30 ;;
  pmc)
    # (HPDDM_HIP_UPLOAD_UNPINNED: under --pmc the copy from the pinned upload ring faulted in round 3)
    cd /tmp && export TMPDIR=/tmp
    export HPDDM_HIP_UPLOAD_UNPINNED=1
    PARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-gmres --no-two-level --no-configs-1 --no-shares --options=-hpddm_hip_numfact_threads=1"
    for cfg in c2 c3; do
      d=$out/pmc_$cfg; mkdir -p $d
      [ $cfg = c2 ] && G="--grid 128" || G=""
      for ctr in FETCH_SIZE WRITE_SIZE; do
        timeout 300 rocprofv3 --pmc $ctr -d $d/raw_$ctr -o p -- python $R/bench.py $G $PARGS > $d/pmc_$ctr.log 2>&1 || { echo "$cfg $ctr pass failed"; tail -3 $d/pmc_$ctr.log; rm -rf $d/raw_$ctr; continue; }
        pdb=$(find $d/raw_$ctr -name "*.db" | head -1)
        python $R/scripts/pmc_summary.py "$pdb" > $d/pmc_$ctr.csv
        python $R/scripts/pmc_total.py "$pdb" 4 > $d/pmc_${ctr}_last_solve.txt
        line $d/pmc_$ctr.log $d/pmc_${ctr}_bench_line.json
        rm -rf $d/raw_$ctr
      done
      cat $d/pmc_*_last_solve.txt 2>/dev/null | grep -v "^#" | head -4
    done ;;
  *) echo "unknown stage $st" ;;
  esac
done
ls -la $out | head -50
