#!/bin/bash
# first stream of the device levels: the library stream (as until this round) or a stream of its own -- the sweeps of configs[1] and configs[2]
cd "$(dirname "$0")/.." || exit 1
for own in 0 1; do
  if [ $own = 1 ]; then export HPDDM_HIP_FACTOR_OWN_STREAM0=1; else unset HPDDM_HIP_FACTOR_OWN_STREAM0; fi
  echo "== own first stream: $own"
  timeout 300 python bench.py --grid 128 --no-two-level --steps 50 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('configs[1]: applies/s', round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'frac', round(d['roofline']['frac'],4))"
  timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-configs-1 --no-shares 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('configs[2]: applies/s', round(d['value'],2), 'ms', round(d['ms_per_step'],3), 'frac', round(d['roofline']['frac'],4), 'setup', d['config']['setup_seconds'], 'geneo', d['two_level']['coarse_space_seconds'])"
done
