#!/bin/bash
# registers, spills, scratch and occupancy of every kernel of one source file of the library (compiler remarks; no GPU needed)
# usage: scripts/kernel_resources.sh hpddm_amd/csrc/sptrsv.hip
src="$1"; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -mavx2 -mfma -fopenmp -I"$(dirname "$src")" -c "$src" -o /dev/null -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c '
import re, sys, subprocess
rows, cur = [], None
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}; rows.append(cur); continue
    for key in ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "VGPRs Spill", "SGPRs Spill", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"):
        m = re.search(re.escape(key) + r": (\d+)", line)
        if m and cur is not None and line.strip().startswith("remark") is False: pass
        if m and cur is not None and (key + ":") in line and key not in cur: cur[key] = int(m.group(1))
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.split("\n") if rows else []
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n).replace("hpddm_hip::", "")
    print("%-70s vgpr %3d agpr %3d scratch %4d vspill %3d occ %d" % (n[:70], r.get("VGPRs", -1), r.get("AGPRs", 0), r.get("ScratchSize [bytes/lane]", 0), r.get("VGPRs Spill", 0), r.get("Occupancy [waves/SIMD]", 0)))
'
