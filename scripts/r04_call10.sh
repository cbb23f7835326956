#!/bin/bash
# device levels of two factorisations in flight (a pool of scratch slots with their own streams): parity of the factorisation paths,
# then the set-up phases of configs[2] for slots x host threads
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_pivoting.py tests/test_complex.py tests/test_helmholtz.py tests/test_gpu_dropin.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r04_call10_tests.log
cat gpurun_out/r04_call10_tests.log
for combo in "1 2 2" "2 2 2" "3 3 3" "2 3 3"; do
  set -- $combo
  echo "== slots $1 numfact threads $2 gevp threads $3"
  HPDDM_HIP_DEVICE_SLOTS=$1 HPDDM_HIP_GEVP_THREADS=$3 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs-1 --no-shares --options=-hpddm_hip_numfact_threads=$2 2>&1 | grep '^{"metric"' > gpurun_out/r04_call10_s$1_t$2_g$3.json
  python - <<PY
import json
d = json.loads(open("gpurun_out/r04_call10_s$1_t$2_g$3.json").read())
print("value", d["value"], "setup", d["config"]["setup_seconds"], d["config"].get("setup_seconds_by_phase_summed_over_subdomains"), "two_level", {k: v for k, v in d["two_level"].items() if "seconds" in k}, "its", d["two_level"]["gmres"]["iterations"], d["one_level"]["gmres"]["iterations"])
PY
done
