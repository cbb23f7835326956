#!/usr/bin/env python3
"""Copy the artefacts of scripts/r06_profiles.sh (gpurun_out/r06/) into profiles/ under their round-6 names and derive
profiles/r06_pmc_traffic_c2.json / r06_pmc_traffic_c3.json (HBM traffic of one batched SpTRSV from the FETCH_SIZE / WRITE_SIZE passes)."""
import json
import os
import shutil

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src, dst = os.path.join(R, "gpurun_out", "r06"), os.path.join(R, "profiles")
names = {"bench_default_stdout.json": "r06_bench_default_stdout.json", "kernel_stats.csv": "r06_bench_c3_kernel_stats.csv",
         "sptrsv_sweeps.csv": "r06_bench_c3_sptrsv_sweeps.csv", "trace_bench_line.json": "r06_bench_c3_trace_bench_line.json",
         "levels_c3.txt": "r06_bench_c3_sptrsv_levels.txt", "levels_c2.txt": "r06_bench_c2_sptrsv_levels.txt", "levels_c4share_helmholtz.txt": "r06_bench_c4share_helmholtz_sptrsv_levels.txt",
         "deflation_256.txt": "r06_deflation_times.txt",
         "bench_c2_stdout.json": "r06_bench_c2_stdout.json", "bench_c4share_helmholtz_stdout.json": "r06_bench_c4share_helmholtz_stdout.json",
         "bench_c3share_elasticity_stdout.json": "r06_bench_c3share_elasticity_stdout.json", "gpu_tests.log": "r06_gpu_tests_final.log",
         "share4_helmholtz.json": "r06_share4_helmholtz_shared_gpu.json", "share8_elasticity.json": "r06_share8_elasticity_shared_gpu.json"}
for a, b in names.items():
    p = os.path.join(src, a)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(dst, b))
        print("profiles/" + b)


def traffic(cfg, grid):
    """(2 FETCH_SIZE + WRITE_SIZE) * 1024 of the last batched SpTRSV of the two PMC runs -> profiles/r06_pmc_traffic_<cfg>.json"""
    d = os.path.join(src, "pmc_" + cfg)
    if not (os.path.exists(os.path.join(d, "pmc_FETCH_SIZE_last_solve.txt")) and os.path.exists(os.path.join(d, "pmc_WRITE_SIZE_last_solve.txt"))):
        return
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        shutil.copy(os.path.join(d, f"pmc_{ctr}.csv"), os.path.join(dst, f"r06_pmc_{ctr.lower()}_{cfg}.csv"))
        shutil.copy(os.path.join(d, f"pmc_{ctr}_last_solve.txt"), os.path.join(dst, f"r06_pmc_{ctr.lower()}_{cfg}_last_solve.txt"))
    fetch = float(open(os.path.join(d, "pmc_FETCH_SIZE_last_solve.txt")).readline().split()[1])
    write = float(open(os.path.join(d, "pmc_WRITE_SIZE_last_solve.txt")).readline().split()[1])
    line = json.load(open(os.path.join(d, "pmc_FETCH_SIZE_bench_line.json")))
    alg = line["roofline"]["bytes_alg_per_sweep"]
    total = (2.0 * fetch + write) * 1024.0
    out = {"config": line["config"]["workload"], "unit": "bytes per batched SpTRSV (forward + backward sweep of the 8 subdomains, all four stream groups)",
           "FETCH_SIZE_KB_per_sweep": fetch, "WRITE_SIZE_KB_per_sweep": write, "gfx950_fetch_correction": 2.0, "traffic_bytes": total,
           "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024  (MI355X_MICROARCH.md, HBM section: on gfx950 FETCH_SIZE reads 1/2 of a wide coalesced 16 B/lane stream)",
           "algorithmic_bytes": alg, "stored_panel_bytes": line["roofline"]["stored_bytes_per_sweep"], "traffic_over_algorithmic": total / alg,
           "collected": "scripts/r06_profiles.sh pmc: separate rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE runs of `HPDDM_HIP_UPLOAD_UNPINNED=1 python bench.py" + grid +
                        " --steps 3 --warmup 1 --no-cpu-baseline --no-gmres --no-two-level --no-configs-1 --no-shares --options=-hpddm_hip_numfact_threads=1`, "
                        "sum over the kernels of the last batched SpTRSV (scripts/pmc_total.py)"}
    json.dump(out, open(os.path.join(dst, f"r06_pmc_traffic_{cfg}.json"), "w"), indent=1)
    print(f"profiles/r06_pmc_traffic_{cfg}.json: x{total / alg:.3f}")


traffic("c2", " --grid 128")
traffic("c3", "")
