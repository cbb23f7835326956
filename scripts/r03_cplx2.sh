#!/bin/bash
mkdir -p gpurun_out/r03
export OMP_NUM_THREADS=8
O=gpurun_out/r03
timeout 900 python -m pytest tests/test_complex.py tests/test_gpu_parity.py tests/test_distributed.py tests/test_gpu_full_size.py tests/test_sptrsv16.py tests/test_gpu_dropin.py -m gpu -x -q > $O/cplx2_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|Error|assert" $O/cplx2_tests.log | head
timeout 300 python bench.py --problem helmholtz --grid 64 --mu 8 --steps 20 --no-shares > $O/helm_v5.log 2>&1; grep '^{"metric"' $O/helm_v5.log | tail -1 | python -c "import json,sys; o=json.load(sys.stdin); print('helm: setup', o['config']['setup_seconds'], 'apply ms', o['ms_per_step'], 'phases', o['phases_ms'], 'defl', o['two_level']['deflation_ms'], o['two_level']['deflation_panel_GBps'], 'bgmres', o['two_level']['gmres']['iterations'], o['two_level']['gmres']['seconds'])"
