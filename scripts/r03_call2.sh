#!/bin/bash
# round 3, GPU call 2: full GPU suite on the pruned library; where the time of the 8-complex-rhs sweep goes (SQ counters);
# the opt-in factorisation paths (big-tile GEMM, sparse upload) under the numfact tests and timed at 129^3
mkdir -p gpurun_out/r03
export OMP_NUM_THREADS=8
O=gpurun_out/r03
timeout 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests_pruned.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests_pruned.log
# opt-in factorisation paths under the tests that factorise
HPDDM_HIP_GEMM=128 HPDDM_HIP_SPARSE_UPLOAD=1 timeout 600 python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_parity.py tests/test_complex.py tests/test_elasticity.py -m gpu -x -q > $O/gpu_tests_optin.log 2>&1; echo "opt-in tests rc=$?"; tail -3 $O/gpu_tests_optin.log
HPDDM_HIP_PROFILE=1 timeout 300 python scripts/time_numfact.py 129 chol device > $O/numfact129_base.log 2>&1; grep -E "numfact|device levels" $O/numfact129_base.log | tail -4
HPDDM_HIP_PROFILE=1 HPDDM_HIP_GEMM=128 HPDDM_HIP_SPARSE_UPLOAD=1 timeout 300 python scripts/time_numfact.py 129 chol device > $O/numfact129_optin.log 2>&1; grep -E "numfact|device levels" $O/numfact129_optin.log | tail -4
cd /tmp && export TMPDIR=/tmp
HPDDM_HIP_GEMM=128 HPDDM_HIP_SPARSE_UPLOAD=1 timeout 400 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/numfact_trace -o t -- python $OLDPWD/scripts/time_numfact.py 129 chol device > $OLDPWD/$O/numfact_trace.log 2>&1
cd $OLDPWD
db=$(find $O/numfact_trace -name "*.db" | head -1); python scripts/prof_summary.py "$db" time_numfact.py 129 chol device > $O/numfact129_kernel_stats.csv; head -14 $O/numfact129_kernel_stats.csv; rm -rf $O/numfact_trace
# the 8-complex-rhs sweep of the Helmholtz share: SQ counters per kernel
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OLDPWD/$O/helm_pmc1 -o p -- python $OLDPWD/bench.py --problem helmholtz --grid 64 --mu 8 --steps 5 --warmup 1 --no-gmres --no-shares > $OLDPWD/$O/helm_pmc1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM -d $OLDPWD/$O/helm_pmc2 -o p -- python $OLDPWD/bench.py --problem helmholtz --grid 64 --mu 8 --steps 5 --warmup 1 --no-gmres --no-shares > $OLDPWD/$O/helm_pmc2.log 2>&1
cd $OLDPWD
for k in 1 2; do db=$(find $O/helm_pmc$k -name "*.db" | head -1); python scripts/pmc_summary.py "$db" > $O/helm_pmc$k.csv; rm -rf $O/helm_pmc$k; done
grep -E "sptrsv_(fwd|bwd)_kernel<16" $O/helm_pmc1.csv | head -40
