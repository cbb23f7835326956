#!/bin/bash
# round 4, second GPU call: the whole GPU suite on the fused-scaling library, hardware-queue / stream-group experiment on the small
# trees, the deflation panel with 8 right-hand sides (fused scaling on / off) and the rocprofv3 evidence of the kernels on the line
cd "$(dirname "$0")/.." || exit 1
R=$PWD
out=$R/gpurun_out/r04
mkdir -p $out
export OMP_NUM_THREADS=8
timeout 1500 python -m pytest tests -q -m gpu > $out/call2_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -15 $out/call2_gpu_tests.log | cut -c1-300
for cfg in "4 4" "8 8" "6 8" "8 4"; do
  set -- $cfg
  for wl in "--grid 128 --no-two-level" "--problem helmholtz --grid 64 --mu 8 --geneo-nu 12"; do
    HPDDM_HIP_STREAMS=$1 GPU_MAX_HW_QUEUES=$2 timeout 300 python bench.py $wl --no-cpu-baseline --no-gmres --no-configs-1 --no-shares 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('streams $1 hwq $2 [$wl]: apply %.3f ms sptrsv %.3f ms frac %.4f' % (d['ms_per_step'], d['phases_ms']['sptrsv'], d['roofline']['frac']))"
  done
done 2>&1 | tee $out/call2_streams_hwq.txt
MUS=1,8 timeout 600 python scripts/time_deflation.py 256 "" "-hpddm_hip_fused_scaling 0" 2>&1 | tee $out/call2_deflation_fused_scaling.txt
cd /tmp && export TMPDIR=/tmp
export MUS=8
timeout 400 rocprofv3 --kernel-trace --stats -d $out/tr_mfma -o t -- python $R/scripts/time_deflation.py 256 > $out/tr_mfma.log 2>&1
db=$(find $out/tr_mfma -name "*.db" | head -1)
python $R/scripts/prof_summary.py "$db" MUS=8 rocprofv3 --kernel-trace --stats -- python scripts/time_deflation.py 256 > $out/call2_deflation_mfma_mu8_kernel_stats.csv
rm -rf $out/tr_mfma
head -12 $out/call2_deflation_mfma_mu8_kernel_stats.csv
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $out/pmc_mfma -o p -- python $R/scripts/time_deflation.py 256 > $out/pmc_mfma.log 2>&1
pdb=$(find $out/pmc_mfma -name "*.db" | head -1)
python $R/scripts/pmc_summary.py "$pdb" > $out/call2_pmc_mfma_deflation.csv
rm -rf $out/pmc_mfma
grep -E "k_zt_mfma|k_z_mfma|k_halo" $out/call2_pmc_mfma_deflation.csv | head -20
cd $R
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $out/tr_h -o t -- python $R/bench.py --problem helmholtz --grid 64 --mu 8 --geneo-nu 12 --no-cpu-baseline --no-configs-1 --no-shares > $out/tr_h.log 2>&1
db=$(find $out/tr_h -name "*.db" | head -1)
python $R/scripts/prof_summary.py "$db" rocprofv3 --kernel-trace --stats -- python bench.py --problem helmholtz --grid 64 --mu 8 --geneo-nu 12 > $out/call2_helmholtz_share_kernel_stats.csv
rm -rf $out/tr_h
head -25 $out/call2_helmholtz_share_kernel_stats.csv
