#!/bin/bash
mkdir -p gpurun_out/r03
export OMP_NUM_THREADS=8
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03/gpu_tests_all.log 2>&1; echo "gpu tests rc=$?"; grep -E "passed|failed|Error" gpurun_out/r03/gpu_tests_all.log | head
timeout 300 python bench.py --problem helmholtz --grid 64 --mu 8 --steps 20 --no-shares > gpurun_out/r03/helm_v4.log 2>&1; grep '^{"metric"' gpurun_out/r03/helm_v4.log | tail -1 | python -c "import json,sys; o=json.load(sys.stdin); print('helm: setup', o['config']['setup_seconds'], 'apply ms', o['ms_per_step'], 'phases', o['phases_ms'], 'defl', o['two_level']['deflation_ms'], 'bgmres', o['two_level']['gmres']['iterations'], o['two_level']['gmres']['seconds'])"
