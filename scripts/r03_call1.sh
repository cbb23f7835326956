#!/bin/bash
# round 3, GPU call 1: the new multi-process tests, the default bench line with the extra shares, and the N>1 layouts on a shared GPU
mkdir -p gpurun_out/r03
export OMP_NUM_THREADS=8
timeout 900 python -m pytest tests/test_distributed.py -m gpu -x -q > gpurun_out/r03/dist_tests.log 2>&1; echo "dist tests rc=$?"
tail -5 gpurun_out/r03/dist_tests.log
BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 8 --problem elasticity --grid 16 --geneo-nu 6 --steps 5 --warmup 2 > gpurun_out/r03/share8_elasticity.json 2> gpurun_out/r03/share8_elasticity.err; echo "share8 rc=$?"
BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 4 --problem helmholtz --grid 16 --mu 8 --steps 5 --warmup 2 > gpurun_out/r03/share4_helmholtz.json 2> gpurun_out/r03/share4_helmholtz.err; echo "share4 rc=$?"
BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 8 --grid 32 --strong --geneo-nu 8 --steps 5 --warmup 2 > gpurun_out/r03/share8_strong.json 2> gpurun_out/r03/share8_strong.err; echo "strong rc=$?"
tail -c 600 gpurun_out/r03/share8_elasticity.json; tail -3 gpurun_out/r03/share8_elasticity.err
tail -c 300 gpurun_out/r03/share4_helmholtz.json; tail -3 gpurun_out/r03/share4_helmholtz.err
tail -c 300 gpurun_out/r03/share8_strong.json; tail -3 gpurun_out/r03/share8_strong.err
timeout 900 python bench.py > gpurun_out/r03/default.json 2> gpurun_out/r03/default.err; echo "default rc=$?"
tail -c 1500 gpurun_out/r03/default.json
