#!/bin/bash
# Round 6: counters and trace of the 8-right-hand-side deflation kernels (k_zt_mfma2 / k_z_mfma2, real panel of configs[2] and the compact
# complex panel of the Helmholtz share) and of the GMV, on the factorisation-free harness scripts/time_deflation.py.
#   gpurun --timeout 900 -- 'bash scripts/r06_mfma.sh'   ->  gpurun_out/r06/{pmc_mfma.csv, pmc_mfma_utilisation.csv, tr_mfma_stats.csv, *_z.*}
cd "$(dirname "$0")/.." || exit 1
R=$PWD
out=$R/gpurun_out/r06
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export MUS=8
for what in real z; do
  [ $what = real ] && ARG=256 || ARG=helmholtz
  sfx=""; [ $what = z ] && sfx="_z"
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $out/pmc_mfma$sfx -o p -- python $R/scripts/time_deflation.py $ARG > $out/pmc_mfma$sfx.log 2>&1
  pdb=$(find $out/pmc_mfma$sfx -name "*.db" | head -1)
  python $R/scripts/pmc_summary.py "$pdb" > $out/pmc_mfma$sfx.csv
  python $R/scripts/mfma_util.py $out/pmc_mfma$sfx.csv > $out/pmc_mfma_utilisation$sfx.csv
  rm -rf $out/pmc_mfma$sfx
  timeout 300 rocprofv3 --kernel-trace --stats -d $out/tr_mfma$sfx -o t -- python $R/scripts/time_deflation.py $ARG > $out/tr_mfma$sfx.log 2>&1
  db=$(find $out/tr_mfma$sfx -name "*.db" | head -1)
  python $R/scripts/prof_summary.py "$db" MUS=8 rocprofv3 --kernel-trace --stats -- python scripts/time_deflation.py $ARG > $out/tr_mfma_stats$sfx.csv
  rm -rf $out/tr_mfma$sfx
  tail -2 $out/tr_mfma$sfx.log; grep -E "mfma" $out/pmc_mfma_utilisation$sfx.csv; head -8 $out/tr_mfma_stats$sfx.csv | cut -c1-160
done
