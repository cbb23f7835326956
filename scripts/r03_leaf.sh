#!/bin/bash
# leaf size of the nested dissection against the sweep time of the Helmholtz share (16-column engine, latency-bound narrow levels)
mkdir -p gpurun_out/r03
for leaf in 32 48 64 96 128; do
  for mu in 8 1; do
    timeout 120 python bench.py --problem helmholtz --grid 64 --mu $mu --leaf $leaf --steps 20 --no-shares --no-gmres > gpurun_out/r03/leaf_${leaf}_mu$mu.log 2>&1
    grep '^{"metric"' gpurun_out/r03/leaf_${leaf}_mu$mu.log | tail -1 | python -c "import json,sys; o=json.load(sys.stdin); print('leaf $leaf mu $mu: apply', round(o['ms_per_step'],3), 'ms sptrsv', round(o['phases_ms']['sptrsv'],3), 'frac', round(o['roofline']['frac'],4), 'nnzL', o['config']['nnz_L_per_gpu'], 'levels', o['config']['levels'], 'setup', o['config']['setup_seconds'])"
  done
done
