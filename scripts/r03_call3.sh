#!/bin/bash
# round 3, GPU call 3: the 16-column MFMA engine -- its tests, the complex tests, the Helmholtz share of configs[4] with per-level tables
mkdir -p gpurun_out/r03
export OMP_NUM_THREADS=8
O=gpurun_out/r03
timeout 600 python -m pytest tests/test_sptrsv16.py tests/test_complex.py -m gpu -x -q > $O/s16_tests.log 2>&1; echo "s16 tests rc=$?"; grep -E "passed|failed|Error|error|assert" $O/s16_tests.log | head -20
timeout 300 python bench.py --problem helmholtz --grid 64 --mu 8 --steps 20 --no-shares > $O/s16_helm.log 2>&1; echo "helm rc=$?"
grep '^{"metric"' $O/s16_helm.log | tail -1 > $O/s16_helm.json
python - <<'PY'
import json
o=json.load(open("gpurun_out/r03/s16_helm.json"))
print("helm mu=8:", o["value"], "applies/s", o["ms_per_step"], "ms; sptrsv", o["phases_ms"]["sptrsv"], "ms frac", o["roofline"]["frac"], "bgmres", o["two_level"]["gmres"])
PY
timeout 200 python scripts/sweep_plan.py --helmholtz 64,64,128 --mu 1,8 --levels --reps 20 "HPDDM_HIP_STREAMS=1" "" > $O/s16_levels_helm.txt 2>&1; grep "^==" $O/s16_levels_helm.txt
timeout 300 python bench.py --grid 128 --no-two-level --mu 16 --steps 20 --no-cpu-baseline --no-gmres > $O/s16_c1_mu16.log 2>&1; echo "c1 mu16 rc=$?"
grep '^{"metric"' $O/s16_c1_mu16.log | tail -1 | python -c "import json,sys; o=json.load(sys.stdin); print('c1 mu=16:', o['ms_per_step'], 'ms per apply; sptrsv', o['phases_ms']['sptrsv'], 'frac', o['roofline']['frac'])"
