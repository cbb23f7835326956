#!/usr/bin/env python3
"""Copy the artefacts of scripts/r04_final.sh (gpurun_out/r04f/) into profiles/ under their round-4 names and derive
profiles/r04_pmc_traffic_c3.json (HBM traffic of one batched SpTRSV from the FETCH_SIZE / WRITE_SIZE passes)."""
import json
import os
import shutil

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src, dst = os.path.join(R, "gpurun_out", "r04f"), os.path.join(R, "profiles")
names = {"bench_default_stdout.json": "r04_bench_default_stdout.json", "kernel_stats.csv": "r04_bench_c3_kernel_stats.csv",
         "sptrsv_sweeps.csv": "r04_bench_c3_sptrsv_sweeps.csv", "trace_bench_line.json": "r04_bench_c3_trace_bench_line.json",
         "pmc_FETCH_SIZE.csv": "r04_pmc_fetch_size_c3.csv", "pmc_WRITE_SIZE.csv": "r04_pmc_write_size_c3.csv",
         "pmc_FETCH_SIZE_last_solve.txt": "r04_pmc_fetch_size_c3_last_solve.txt", "pmc_WRITE_SIZE_last_solve.txt": "r04_pmc_write_size_c3_last_solve.txt",
         "levels_c3.txt": "r04_bench_c3_sptrsv_levels.txt", "levels_c2.txt": "r04_bench_c2_sptrsv_levels.txt", "levels_c4share_helmholtz.txt": "r04_bench_c4share_helmholtz_sptrsv_levels.txt",
         "bench_c2_stdout.json": "r04_bench_c2_stdout.json", "bench_c4share_helmholtz_stdout.json": "r04_bench_c4share_helmholtz_stdout.json",
         "bench_c3share_elasticity_stdout.json": "r04_bench_c3share_elasticity_stdout.json", "gpu_tests_final.log": "r04_gpu_tests_final.log",
         "share4_helmholtz.json": "r04_share4_helmholtz_shared_gpu.json", "share8_elasticity.json": "r04_share8_elasticity_shared_gpu.json"}
for a, b in names.items():
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, b))


def traffic(srcdir, target, how):
    """(2 FETCH_SIZE + WRITE_SIZE) * 1024 of the last batched SpTRSV of the two PMC runs under srcdir -> profiles/<target>"""
    fetch = float(open(os.path.join(srcdir, "pmc_FETCH_SIZE_last_solve.txt")).readline().split()[1])
    write = float(open(os.path.join(srcdir, "pmc_WRITE_SIZE_last_solve.txt")).readline().split()[1])
    line = json.load(open(os.path.join(srcdir, "pmc_FETCH_SIZE_bench_line.json")))
    alg = line["roofline"]["bytes_alg_per_sweep"]
    total = (2.0 * fetch + write) * 1024.0
    out = {"config": line["config"]["workload"], "unit": "bytes per batched SpTRSV (forward + backward sweep of the 8 subdomains, all four stream groups)",
           "FETCH_SIZE_KB_per_sweep": fetch, "WRITE_SIZE_KB_per_sweep": write, "gfx950_fetch_correction": 2.0, "traffic_bytes": total,
           "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024  (MI355X_MICROARCH.md, HBM section: on gfx950 FETCH_SIZE reads 1/2 of a wide coalesced 16 B/lane stream)",
           "algorithmic_bytes": alg, "stored_panel_bytes": line["roofline"]["stored_bytes_per_sweep"], "traffic_over_algorithmic": total / alg, "collected": how}
    json.dump(out, open(os.path.join(dst, target), "w"), indent=1)
    print(json.dumps(out, indent=1))


if os.path.exists(os.path.join(src, "pmc_FETCH_SIZE_last_solve.txt")):
    traffic(src, "r04_pmc_traffic_c3.json", "scripts/r04_final.sh: separate rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE runs of `HPDDM_HIP_UPLOAD_UNPINNED=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline "
            "--no-gmres --no-two-level --no-configs-1 --no-shares --options=-hpddm_hip_numfact_threads=1`, sum over the kernels of the last batched SpTRSV (scripts/pmc_total.py)")
c2 = os.path.join(R, "gpurun_out", "r04f", "c2")
if os.path.exists(os.path.join(c2, "pmc_FETCH_SIZE_last_solve.txt")):
    for a, b in {"pmc_FETCH_SIZE.csv": "r04_pmc_fetch_size_c2.csv", "pmc_WRITE_SIZE.csv": "r04_pmc_write_size_c2.csv",
                 "pmc_FETCH_SIZE_last_solve.txt": "r04_pmc_fetch_size_c2_last_solve.txt", "pmc_WRITE_SIZE_last_solve.txt": "r04_pmc_write_size_c2_last_solve.txt"}.items():
        shutil.copy(os.path.join(c2, a), os.path.join(dst, b))
    traffic(c2, "r04_pmc_traffic_c2.json", "scripts/r04_final.sh: separate rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE runs of `HPDDM_HIP_UPLOAD_UNPINNED=1 python bench.py --grid 128 --steps 3 --warmup 1 "
            "--no-cpu-baseline --no-gmres --no-two-level --no-configs-1 --no-shares --options=-hpddm_hip_numfact_threads=1`, sum over the kernels of the last batched SpTRSV (scripts/pmc_total.py)")
