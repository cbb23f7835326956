#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
R=$PWD
out=$R/gpurun_out/r03
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export MUS=8
timeout 300 rocprofv3 --kernel-trace --stats -d $out/tr_mfma -o t -- python $R/scripts/time_deflation.py 256 > $out/tr_mfma.log 2>&1
db=$(find $out/tr_mfma -name "*.db" | head -1)
python $R/scripts/prof_summary.py "$db" rocprofv3 --kernel-trace --stats -- python scripts/time_deflation.py 256 > $out/tr_mfma_stats.csv
rm -rf $out/tr_mfma
head -14 $out/tr_mfma_stats.csv
