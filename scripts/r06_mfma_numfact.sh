#!/bin/bash
# Round 6: MFMA-busy counters of the kernels of one 129^3 numerical factorisation (the f64 products of the device levels, k_gemm_big at
# two workgroups per CU) -- rocprofv3 --pmc on scripts/time_setup_threads.py 129 (uploads through pageable memory under the counters).
#   gpurun --timeout 900 -- 'bash scripts/r06_mfma_numfact.sh'  ->  gpurun_out/r06/pmc_mfma_numfact{.csv,_utilisation.csv}
cd "$(dirname "$0")/.." || exit 1
R=$PWD
out=$R/gpurun_out/r06
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
HPDDM_HIP_UPLOAD_UNPINNED=1 timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $out/pmc_nf -o p -- python $R/scripts/time_setup_threads.py 129 > $out/pmc_mfma_numfact.log 2>&1
pdb=$(find $out/pmc_nf -name "*.db" | head -1)
python $R/scripts/pmc_summary.py "$pdb" > $out/pmc_mfma_numfact.csv
python $R/scripts/mfma_util.py $out/pmc_mfma_numfact.csv > $out/pmc_mfma_numfact_utilisation.csv
rm -rf $out/pmc_nf
tail -2 $out/pmc_mfma_numfact.log; grep -E "gemm|potf2" $out/pmc_mfma_numfact_utilisation.csv | cut -c1-200
