#!/bin/bash
# developer aid: the sweeps of two builds on one box -- this tree against a library kept under _abl/<name> (hpddm_amd/*.py + libhpddm_hip.so + scripts/sweep_plan.py)
cd "$(dirname "$0")/.." || exit 1
R=$PWD; out=$R/gpurun_out/r05ab3; rm -rf $out; mkdir -p $out
export OMP_NUM_THREADS=8
for t in base CONTIG base CONTIG; do
  [ $t = base ] && cd $R || cd $R/_abl/$t
  timeout 600 python scripts/sweep_plan.py --grid 256 --levels --reps 10 "HPDDM_HIP_STREAMS=1" > $out/${t}_$RANDOM.txt 2>&1
done
grep "^==" $out/*.txt
