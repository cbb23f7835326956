#!/bin/bash
# the streams of the sweep groups picked by timing: (1) configs[1] whatever streams were created ahead (the deals that cost 25 % before),
# (2) parity of the operator paths, (3) the default bench line
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r04h
rm -rf $out && mkdir -p $out
run() {
  timeout 300 python bench.py --grid 128 --no-two-level --steps 50 --no-cpu-baseline --no-gmres 2>$out/err.txt | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  configs[1]: applies/s', round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'frac', round(d['roofline']['frac'],4))"
  grep "build_plans" $out/err.txt | tail -1
}
export HPDDM_HIP_PROFILE=1
for pat in 0,0,0 1,0,0 2,0,0 3,0,0 0,2,3; do echo "== unused streams ahead of the groups: $pat"; HPDDM_HIP_STREAM_PATTERN=$pat run; done
unset HPDDM_HIP_PROFILE
echo "== tuning off, 1,0,0"; HPDDM_HIP_STREAM_PATTERN=1,0,0 timeout 300 python bench.py --grid 128 --no-two-level --steps 50 --no-cpu-baseline --no-gmres --options=-hpddm_hip_tune_streams=0 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  configs[1]: applies/s', round(d['value'],1), 'ms', round(d['ms_per_step'],3))"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py tests/test_helmholtz.py -q -m gpu -x 2>&1 | tail -3
export OMP_NUM_THREADS=8
( time timeout 1200 python bench.py ) > $out/bench_default_stdout.log 2> $out/bench_default_stderr.log
grep '^{"metric"' $out/bench_default_stdout.log | tail -1 > $out/bench_default_stdout.json
python - <<PY
import json
d = json.loads(open("$out/bench_default_stdout.json").read())
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "setup", d["config"]["setup_seconds"], {k: v for k, v in d["two_level"].items() if "seconds" in k}, "cpu", d["cpu_baseline"]["value"], "c1", d["configs_1"]["roofline"]["frac"], d["configs_1"]["apply_ms"], "c3", d["configs_3_share"]["apply_ms"], "c4", d["configs_4_share"]["apply_ms"])
PY
