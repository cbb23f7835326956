#!/bin/bash
# MFMA utilisation of the deflation panel (k_zt_mfma / k_z_mfma: 8 and 16 right-hand sides, nu = 20, configs[2] sizes) from the SQ
# counters, on the factorisation-free harness scripts/time_deflation.py
cd "$(dirname "$0")/.." || exit 1
R=$PWD
out=$R/gpurun_out/r03
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export MUS=8,16
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $out/pmc_mfma -o p -- python $R/scripts/time_deflation.py 256 > $out/pmc_mfma.log 2>&1
pdb=$(find $out/pmc_mfma -name "*.db" | head -1)
python $R/scripts/pmc_summary.py "$pdb" > $out/pmc_mfma.csv
rm -rf $out/pmc_mfma
tail -3 $out/pmc_mfma.log; grep -E "k_zt_mfma|k_z_mfma" $out/pmc_mfma.csv
