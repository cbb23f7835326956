#!/bin/bash
# Round 6: the 16-column engine on the Helmholtz share of configs[4] -- correctness (tests of the engine) + per-level table, one stream and four
#   gpurun --timeout 900 -- 'bash scripts/r06_e16.sh TAG "cfg1" "cfg2" ...'   ->  gpurun_out/r06_e16_TAG.{log,txt}
cd "$(dirname "$0")/.." || exit 1
tag=${1:-x}; shift
timeout 600 python -m pytest tests/test_sptrsv16.py -x -q -m gpu > gpurun_out/r06_e16_$tag.log 2>&1
tail -3 gpurun_out/r06_e16_$tag.log
timeout 400 python scripts/sweep_plan.py --helmholtz 64,64,128 --mu 8 --levels "$@" > gpurun_out/r06_e16_$tag.txt 2>&1
grep -E "^==|total|level 900|rror" gpurun_out/r06_e16_$tag.txt
