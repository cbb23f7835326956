#!/bin/bash
# developer aid: the sweeps of two builds on one box -- this tree against the library kept in _abl/OLD (hpddm_amd/*.py + libhpddm_hip.so + scripts/sweep_plan.py)
cd "$(dirname "$0")/.." || exit 1
R=$PWD; out=$R/gpurun_out/r05ab2; rm -rf $out; mkdir -p $out
export OMP_NUM_THREADS=8
timeout 600 python scripts/check_sptrsv.py > $out/check.txt 2>&1; echo "check exit $?"; tail -1 $out/check.txt
for t in OLD new OLD new; do
  [ $t = new ] && cd $R || cd $R/_abl/$t
  timeout 300 python scripts/sweep_plan.py --grid 128 --levels --reps 30 "HPDDM_HIP_STREAMS=1" "" >> $out/${t}_c2.txt 2>&1
  timeout 600 python scripts/sweep_plan.py --grid 256 --levels --reps 10 "HPDDM_HIP_STREAMS=1" "" >> $out/${t}_c3.txt 2>&1
  grep "^==" $out/${t}_c2.txt $out/${t}_c3.txt | tail -4
done
