#!/bin/bash
# one upload per front (children's blocks, entry lists, row maps), the tile products in place: residuals of the three kinds, parity
# of the device levels (real and complex), the numerical phase of a 129^3 subdomain, the set-up phases of configs[2]
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 300 python scripts/time_numfact.py 65 chol,ldlt,lu device 2>&1 | tail -4
HPDDM_HIP_PROFILE=1 timeout 300 python scripts/time_numfact.py 129 chol device 2>&1 | grep "device levels\|N=129" | tail -8
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_pivoting.py tests/test_complex.py tests/test_helmholtz.py -m gpu -x -q 2>&1 | tail -5
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs-1 --no-shares > gpurun_out/r04_call13_bench.log 2>&1
echo "rc $?"; grep -v '^{"metric"' gpurun_out/r04_call13_bench.log | tail -3
python - <<PY
import json
rows = [l for l in open("gpurun_out/r04_call13_bench.log") if l.startswith('{"metric"')]
if rows:
    d = json.loads(rows[0])
    print("value", d["value"], "setup", d["config"]["setup_seconds"], d["config"].get("setup_seconds_by_phase_summed_over_subdomains"), "two_level", {k: v for k, v in d["two_level"].items() if "seconds" in k}, "its", d["two_level"]["gmres"]["iterations"], d["one_level"]["gmres"]["iterations"])
PY
