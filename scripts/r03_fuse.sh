#!/bin/bash
# timing experiment (ABLATION build): bottom levels of the forward sweep in ONE launch, dependencies ignored (wrong results) --
# what a dependency-driven fused sweep could reach at best
# (needs the HPDDM_HIP_FUSE_UNSAFE switch of the ablation build as of commit 6615c05 .. bf48531: removed from the source in round 4)
mkdir -p gpurun_out/r03
timeout 200 python scripts/sweep_plan.py --helmholtz 64,64,128 --mu 1,8 --levels --reps 20 "HPDDM_HIP_STREAMS=1" "HPDDM_HIP_STREAMS=1 HPDDM_HIP_FUSE_UNSAFE=4" "HPDDM_HIP_STREAMS=1 HPDDM_HIP_FUSE_UNSAFE=99" > gpurun_out/r03/fuse_helm.txt 2>&1
grep -E "^==|fwd total|bwd total" gpurun_out/r03/fuse_helm.txt
timeout 200 python scripts/sweep_plan.py --grid 128 --mu 1 --levels --reps 20 "HPDDM_HIP_STREAMS=1" "HPDDM_HIP_STREAMS=1 HPDDM_HIP_FUSE_UNSAFE=6" > gpurun_out/r03/fuse_c2.txt 2>&1
grep -E "^==|fwd total|bwd total" gpurun_out/r03/fuse_c2.txt
