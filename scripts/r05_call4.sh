#!/bin/bash
# round 5, fourth GPU call: complex BCG / BFBCG on the device, compact-Z direct deflation kernels, parallel reduce + coarse solve,
# GMV with the right-hand sides four at a time, the complex CPU leg of the Helmholtz share
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r05d
rm -rf $out && mkdir -p $out
export OMP_NUM_THREADS=8
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_complex.py tests/test_helmholtz.py tests/test_elasticity.py tests/test_gpu_edge_cases.py -q -m gpu > $out/gpu_tests_subset.log 2>&1; echo "gpu tests rc=$?"; tail -25 $out/gpu_tests_subset.log | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_full_size.py -q -m gpu -k "helmholtz or config4 or configs_4 or share" > $out/gpu_tests_full_size.log 2>&1; echo "full size rc=$?"; tail -8 $out/gpu_tests_full_size.log | cut -c1-300
timeout 400 python scripts/time_deflation.py 256 "" > $out/deflation_256.txt 2>&1; tail -4 $out/deflation_256.txt
timeout 200 python scripts/time_deflation.py helmholtz "" > $out/deflation_helmholtz.txt 2>&1; tail -3 $out/deflation_helmholtz.txt
timeout 500 python bench.py --problem helmholtz --grid 64 --mu 8 --geneo-nu 12 --steps 20 > $out/bench_c4share.log 2>&1
grep '^{"metric"' $out/bench_c4share.log | tail -1 > $out/bench_c4share.json
python - <<PY
import json
d = json.loads(open("$out/bench_c4share.json").read())
print("c4 share: value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "two_level", {k: v for k, v in d.get("two_level", {}).items() if "ms" in k or "seconds" in k}, "cpu", {k: d.get("cpu_baseline", {}).get(k) for k in ("value", "cores", "gpu_one_level_over_cpu")})
print(d.get("cpu_baseline", {}).get("sample"))
PY
