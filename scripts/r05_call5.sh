#!/bin/bash
# round 5, fifth GPU call: what the memory system gives the backward tiles' access pattern (128-column tiles strided by the row
# pitch against 512-column tiles), the helmholtz test that failed, GMV / deflation with 8 right-hand sides
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r05e
rm -rf $out && mkdir -p $out
export OMP_NUM_THREADS=8
timeout 120 scripts/micro/coldot_probe > $out/coldot_probe.txt 2>&1; cat $out/coldot_probe.txt
timeout 300 python -m pytest tests/test_helmholtz.py -q -m gpu 2>&1 | tail -3
MUS=1,8 timeout 400 python scripts/time_deflation.py 256 "" > $out/deflation_256.txt 2>&1; tail -3 $out/deflation_256.txt
