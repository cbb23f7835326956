#!/bin/bash
# device factorisation: the factorising tests, then timings (129^3 Cholesky, 65^3 all kinds), kernel stats of one 129^3 factorisation
mkdir -p gpurun_out/r03
export OMP_NUM_THREADS=8
O=gpurun_out/r03
timeout 900 python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_parity.py tests/test_complex.py tests/test_elasticity.py tests/test_sptrsv16.py -m gpu -x -q > $O/numfact_tests.log 2>&1; echo "numfact tests rc=$?"; grep -E "passed|failed|Error" $O/numfact_tests.log | head
HPDDM_HIP_PROFILE=1 timeout 300 python scripts/time_numfact.py 129 chol device > $O/numfact129_v2.log 2>&1; grep -E "device levels|numfact " $O/numfact129_v2.log | tail -3
timeout 300 python scripts/time_numfact.py 65 chol,ldlt,lu device > $O/numfact65_v2.log 2>&1; grep -E "numfact " $O/numfact65_v2.log
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/numfact_trace -o t -- python $OLDPWD/scripts/time_numfact.py 129 chol device > $OLDPWD/$O/numfact_trace.log 2>&1
cd $OLDPWD
db=$(find $O/numfact_trace -name "*.db" | head -1); python scripts/prof_summary.py "$db" time_numfact.py 129 chol device > $O/numfact129_v2_kernel_stats.csv; head -14 $O/numfact129_v2_kernel_stats.csv | cut -c1-200; rm -rf $O/numfact_trace
