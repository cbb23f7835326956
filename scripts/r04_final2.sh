#!/bin/bash
# end of round 4, after the work on the device levels and the complex recycling methods: the whole GPU suite and the default bench line
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r04g
rm -rf $out && mkdir -p $out
export OMP_NUM_THREADS=8
timeout 1500 python -m pytest tests -q -m gpu > $out/gpu_tests_final.log 2>&1; echo "gpu tests rc=$?"; tail -4 $out/gpu_tests_final.log | cut -c1-300
( time timeout 1200 python bench.py ) > $out/bench_default_stdout.log 2> $out/bench_default_stderr.log
grep '^{"metric"' $out/bench_default_stdout.log | tail -1 > $out/bench_default_stdout.json
tail -5 $out/bench_default_stderr.log
python - <<PY
import json
d = json.loads(open("$out/bench_default_stdout.json").read())
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "setup", d["config"]["setup_seconds"], {k: v for k, v in d["two_level"].items() if "seconds" in k}, "cpu", d["cpu_baseline"]["value"], "c1", d["configs_1"]["roofline"]["frac"] if "roofline" in d["configs_1"] else None)
PY
