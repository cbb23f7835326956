#!/bin/bash
mkdir -p gpurun_out/r03
export OMP_NUM_THREADS=8
O=gpurun_out/r03
timeout 900 python -m pytest tests/test_complex.py tests/test_gpu_edge_cases.py tests/test_sptrsv16.py -m gpu -x -q > $O/cplx_tests.log 2>&1; echo "complex + edge tests rc=$?"; grep -E "passed|failed|Error|assert" $O/cplx_tests.log | head
timeout 300 python scripts/time_numfact.py 65 chol,ldlt,lu device > $O/numfact65_v3.log 2>&1; grep -E "numfact " $O/numfact65_v3.log
timeout 300 python bench.py --problem helmholtz --grid 64 --mu 8 --steps 20 --no-shares > $O/helm_v3.log 2>&1; grep '^{"metric"' $O/helm_v3.log | tail -1 | python -c "import json,sys; o=json.load(sys.stdin); print('helm: setup', o['config']['setup_seconds'], 'apply ms', o['ms_per_step'], 'sptrsv', o['phases_ms']['sptrsv'], 'bgmres', o['two_level']['gmres']['iterations'])"
