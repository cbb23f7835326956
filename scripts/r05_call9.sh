#!/bin/bash
# round 5, ninth GPU call: what bounds the launch of the condensed leaves -- instruction counts and busy cycles of the level-0 kernels
cd "$(dirname "$0")/.." || exit 1
R=$PWD
out=$R/gpurun_out/r05i
rm -rf $out && mkdir -p $out
export OMP_NUM_THREADS=8
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $out/counters.txt 2>&1
pass() { # name, env, counters...
  local nm=$1 ev=$2; shift 2
  env $ev HPDDM_HIP_STREAMS=1 timeout 300 rocprofv3 --pmc "$@" -d $out/pmc_$nm -o p -- python $R/scripts/sweep_plan.py --grid 128 --reps 3 "" > $out/pmc_$nm.log 2>&1
  pdb=$(find $out/pmc_$nm -name "*.db" | head -1)
  python $R/scripts/pmc_summary.py "$pdb" | grep "true>\|kernel,counter" > $out/pmc_$nm.csv
  rm -rf $out/pmc_$nm
}
pass insts_pairs "A=1" SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass insts_single "HPDDM_HIP_LEAF_PAIRS=0" SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass busy_pairs "A=1" SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
pass busy_single "HPDDM_HIP_LEAF_PAIRS=0" SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
cat $out/pmc_*.csv | cut -c1-200
tail -3 $out/pmc_busy_single.log
