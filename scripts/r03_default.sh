#!/bin/bash
mkdir -p gpurun_out/r03
timeout 1200 python bench.py > gpurun_out/r03/default2.json 2> gpurun_out/r03/default2.err; echo "default rc=$?"
python - <<'PY'
import json
o=json.loads([l for l in open("gpurun_out/r03/default2.json") if l.startswith("{")][-1])
print("value", o["value"], "ms", o["ms_per_step"], "frac", o["roofline"]["frac"], "setup", o["config"]["setup_seconds"], "geneo", o["two_level"]["coarse_space_seconds"], "coarse", o["two_level"]["coarse_setup_seconds"])
for k in ("configs_1","configs_3_share","configs_4_share"):
    c=o[k]; print(k, c.get("value", c.get("applies_per_sec")), c.get("ms_per_step", c.get("apply_ms")), c["roofline"]["frac"], c.get("setup_seconds"))
print("cpu", o["cpu_baseline"]["value"], o["cpu_baseline"]["cores"])
PY
