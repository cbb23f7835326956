#!/bin/bash
# two scratch slots: where does configs[2] fail?  (stderr kept)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
for g in 128 256; do
  echo "== grid $g"
  HPDDM_HIP_DEVICE_SLOTS=2 timeout 600 python bench.py --grid $g --steps 5 --warmup 2 --no-cpu-baseline --no-configs-1 --no-shares > gpurun_out/r04_call11_g$g.log 2>&1
  echo "rc $?"; tail -c 1500 gpurun_out/r04_call11_g$g.log | grep -v '^{"metric"' | tail -12
done
