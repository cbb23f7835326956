#!/bin/bash
# developer aid (round 6): library variants built with -DHPDDM_BUSH_PF / -DHPDDM_BUSH_OCC (ring depth / wavefronts per SIMD of the bush kernels) on the Helmholtz share
cd "$(dirname "$0")/.." || exit 1
cp hpddm_amd/libhpddm_hip.so /tmp/lib_keep.so
for v in pf4:40 pf6:40 pf8:53 pf8:40; do
  lib=${v%%:*}; lds=${v##*:}
  cp hpddm_amd/csrc/build/lib_$lib.so hpddm_amd/libhpddm_hip.so
  echo "## $lib, HPDDM_HIP_BUSH_LDS=$lds"
  timeout 300 python scripts/sweep_plan.py --helmholtz 64,64,128 --mu 8 --levels "HPDDM_HIP_STREAMS=1 HPDDM_HIP_BUSH_LDS=$lds" "HPDDM_HIP_BUSH_LDS=$lds" 2>&1 | grep -E "^==|level 900|total"
done
cp /tmp/lib_keep.so hpddm_amd/libhpddm_hip.so
