#!/usr/bin/env python3
"""bench.py -- RAS preconditioner applies/s on MI355X (BASELINE.json metric), config 2 of BASELINE.json:
3-D Poisson 128^3, 8 subdomains on one GPU, one-level RAS, HIP local SpTRSV.

A "step" = one apply of the preconditioner,  out = sum_i R_i^T D_i A_i^{-1} R_i in,  for all 8 subdomains of the GPU
(8 level-scheduled SpTRSV + fused D-scaling/halo sum), vectors resident in HBM.  With N GPUs every rank owns its own
128^3 block of 8 subdomains (weak scaling, replicas: the cross-GPU halo is not built this round), and
`value` is the aggregate number of 8-subdomain applies per second.

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (batched SpTRSV against HBM peak, algorithmic bytes of
SURVEY 8(d)), "cpu_baseline" (oracle substitution on the same factors, one host thread per subdomain like the
reference's one-rank-per-subdomain layout), "gmres" (iterations/s of the device-resident GMRES on the same operator).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--grid", dest="n", type=int, default=128, help="global grid is grid^3 cells per GPU")
    ap.add_argument("--subdomains", type=int, default=8)
    ap.add_argument("--mu", type=int, default=1)
    ap.add_argument("--leaf", type=int, default=0, help="nested-dissection leaf size (0 = library default)")
    ap.add_argument("--replicas", action="store_true", help="N>1: independent 8-subdomain blocks per GPU instead of one global problem with a cross-GPU halo")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gmres", action="store_true")
    ap.add_argument("--bgmres", type=int, default=0, metavar="MU", help="extra leg: Block GMRES on MU consistent random right-hand sides (configs[4] solves 8 at a time)")
    ap.add_argument("--no-two-level", action="store_true")
    ap.add_argument("--geneo-nu", type=int, default=20, help="deflation vectors per subdomain of the two-level leg")
    ap.add_argument("--problem", choices=("poisson", "elasticity"), default="poisson",
                    help="poisson: 7-point Laplacian, grid^3 cells per GPU (configs[1], configs[2]); elasticity: trilinear hexahedra, 3 dofs per "
                         "node, grid^3 nodes per GPU (configs[3] is --problem elasticity --grid 64 on 8 GPUs)")
    ap.add_argument("--geneo", action="store_true", help="two-level leg: compute the real GenEO vectors (solveGEVP) instead of the polynomial stand-ins")
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    # BENCH_SHARE_GPU=1 (development only): all ranks use GPU 0 and rendezvous over gloo, to exercise the multi-process
    # code path on a single-GPU box
    share = os.environ.get("BENCH_SHARE_GPU") == "1"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            local = 0
        torch.cuda.set_device(local)
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist = None
        local = 0
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local)
    cpu_coll = share and world > 1

    from hpddm_amd import _lib, hpddm
    from hpddm_amd.generate import generate3d, generate_elasticity3d

    def generate(dims, parts, **kw):
        if args.problem == "elasticity":
            kw.pop("rhs", None)
            kw.setdefault("normalize", True)
            return generate_elasticity3d(dims, parts, overlap=1, sym=True, **kw)
        return generate3d(dims, parts, overlap=1, sym=True, **kw)

    hpddm.require_device()
    _lib.check(_lib.load().HpddmHipSetDevice(dev.index))

    # ---- build the operator (one-time: generator, analysis, factorisation, upload) ----
    t0 = time.time()
    want_cpu = (rank == 0 and world == 1 and not args.no_cpu_baseline)
    opts = "-hpddm_operator_spd" + (" -hpddm_keep_plain 1" if want_cpu else "") + (f" -hpddm_leaf_size {args.leaf}" if args.leaf else "")
    sharded = world > 1 and not args.replicas
    if sharded:
        # ONE global problem: grid^2 x (grid * N) cells, 2 x 2 x 2N boxes; rank r owns the 8 subdomains of its z-slab and
        # exchanges the halo of the two slab faces with ranks r-1 / r+1 (RCCL point-to-point over xGMI)
        assert args.subdomains == 8
        parts = 8 * world
        subs = generate((args.n, args.n, args.n * world), parts, rhs="smooth", grid=(2, 2, 2 * world), first=8 * rank, count=8, normalize=True, neumann=args.geneo)
        A, d = hpddm.schwarz_from_subdomains(subs, first_global=8 * rank, nglobal=parts, options=opts, multiplicity=False,
                                             partition=(rank, [8 * r for r in range(world + 1)]))
        A.enable_distributed(dist, dev, mu_cap=max(1, args.mu, 0 if args.no_two_level else args.geneo_nu), host_staging=cpu_coll)
    else:
        subs = generate(args.n, args.subdomains, rhs="smooth", neumann=args.geneo)
        A, d = hpddm.schwarz_from_subdomains(subs, options=opts, multiplicity=args.problem != "elasticity")
    A.call_numfact()
    t_setup = time.time() - t0
    st = A.stats()
    ntot, mu = int(st["n"]), args.mu

    x = torch.ones(ntot * mu, dtype=torch.float64, device=dev)
    y = torch.zeros_like(x)
    torch.cuda.synchronize()

    def step():
        A.apply_device(x.data_ptr(), y.data_ptr(), mu)

    for _ in range(args.warmup):
        step()
    A.synchronize()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    A.synchronize()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if cpu_coll else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = world * args.steps / elapsed  # every rank applies its own 8-subdomain operator once per step

    out = {
        "metric": "ras_precond_applies_per_sec", "value": value, "unit": "applies/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": (f"BASELINE.json configs[{1 if args.n == 128 else (2 if args.n == 256 else '1-like')}]: 3-D Poisson {args.n}^3 per GPU, "
                                if args.problem == "poisson" else
                                f"BASELINE.json configs[3]-like: 3-D linear elasticity (block-3 CSR), {args.n}^3 nodes per GPU, ") +
                               f"{args.subdomains} subdomains per GPU, one-level RAS (two-level leg reported under 'two_level'), "
                               f"HIP level-scheduled SpTRSV, overlap 1, mu={mu}",
                   "parallelism": ("1 GPU, 8 subdomains batched" if world == 1 else
                                   (f"{world} GPUs, one global {args.n}x{args.n}x{args.n * world} problem, 8 subdomains per GPU, cross-GPU halo by RCCL send/recv" if sharded
                                    else "replicas (one independent 8-subdomain block per GPU)")),
                   "n_dof_per_gpu": ntot, "nnz_L_per_gpu": st["nnz_L"], "levels": st["levels"], "launches_per_sptrsv": st["launches"],
                   "setup_seconds": round(t_setup, 2)},
    }
    gm = None
    if sharded and not args.no_gmres:
        # GMRES on the global problem: every rank takes part (halo + all-reduce inside)
        fb = torch.from_numpy(np.concatenate([s["f"] for s in subs])).to(dev)
        xs = torch.zeros_like(fb)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        it = A.solve_device(fb.data_ptr(), xs.data_ptr(), 1)
        torch.cuda.synchronize()
        tg = time.perf_counter() - t0
        gm = {"iterations": it, "seconds": tg, "iters_per_sec": it / tg, "tol": 1e-6}
    ph = None
    if sharded:
        # collective: every rank runs the same calls
        ph = {"exchange": A.time("exchange", mu=mu, reps=10) * 1e3, "gmv": A.time("gmv", mu=mu, reps=10) * 1e3}
    tl = None
    if sharded and not args.no_two_level:
        # the coarse operator spans the ranks (assembly through the halo transport, coarse gather = all-reduce): collective
        tl = two_level(A, subs, args, np, mu, 10)
    if rank == 0:
        if gm:
            out["gmres"] = gm
        # ---- roofline of the dominant kernel pair (batched SpTRSV), HIP events on the library stream ----
        reps = max(5, min(50, args.steps))
        t_solve = A.time("solve", mu=mu, warmup=2, reps=reps)
        bytes_alg = 2.0 * st["nnz_L"] * 8.0 + 4.0 * st["n"] * mu * 8.0   # 2*nnz(L)*sizeof(K) + 4*n*mu*sizeof(K)
        achieved = bytes_alg / t_solve / 1e9
        out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": None,
                           "kernel": "sptrsv_fwd_kernel + sptrsv_bwd_kernel (one batched forward+backward sweep = %d launches)" % int(st["launches"]),
                           "bytes_alg_per_sweep": bytes_alg, "seconds_per_sweep": t_solve,
                           "stored_bytes_per_sweep": 2.0 * st["stored"] * 8.0}
        # HBM traffic of the same sweep pair from the committed PMC passes (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
        # runs of this command, scripts/final_profiles.sh): only quoted for the workload it was collected on
        pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_traffic.json")
        if args.n == 128 and args.subdomains == 8 and mu == 1 and os.path.exists(pmc):
            with open(pmc) as fh:
                tr = json.load(fh)
            if abs(tr.get("algorithmic_bytes", 0.0) - bytes_alg) < 1e-6 * bytes_alg:
                out["roofline"]["traffic"] = tr["traffic_bytes"]
                out["roofline"]["traffic_source"] = "profiles/r01_pmc_traffic.json ((2*FETCH_SIZE + WRITE_SIZE)*1024, rocprofv3 --pmc, separate passes)"
        out["phases_ms"] = {"sptrsv": t_solve * 1e3}
        if not sharded:  # exchange / GMV of a sharded operator are collective: timed on all ranks below
            out["phases_ms"].update({"exchange": A.time("exchange", mu=mu, reps=reps) * 1e3, "gmv": A.time("gmv", mu=mu, reps=reps) * 1e3})
        elif ph:
            out["phases_ms"].update(ph)
        if not args.no_gmres and world == 1:
            f = [s["f"] for s in subs]
            fb = torch.from_numpy(np.concatenate(f)).to(dev)
            xs = torch.zeros_like(fb)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            it = A.solve_device(fb.data_ptr(), xs.data_ptr(), 1)
            torch.cuda.synchronize()
            tg = time.perf_counter() - t0
            out["gmres"] = {"iterations": it, "seconds": tg, "iters_per_sec": it / tg, "tol": 1e-6}
        if args.bgmres > 1 and world == 1:
            # Block GMRES (IterativeMethod::BGMRES) on args.bgmres right-hand sides made consistent by one exchange
            rng = np.random.default_rng(1)
            rhs = A.exchange([rng.random((s["n"], args.bgmres)) for s in subs])
            flat, _ = A.pack(rhs)
            fb = torch.from_numpy(flat).to(dev)
            xs = torch.zeros_like(fb)
            A.option_parse("-hpddm_krylov_method bgmres")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            it = A.solve_device(fb.data_ptr(), xs.data_ptr(), args.bgmres)
            torch.cuda.synchronize()
            tg = time.perf_counter() - t0
            A.option_parse("-hpddm_krylov_method gmres")
            out["bgmres"] = {"rhs": args.bgmres, "iterations": it, "seconds": tg, "iters_per_sec": it / tg, "rhs_iters_per_sec": it * args.bgmres / tg, "tol": 1e-6}
        if not args.no_two_level and world == 1:
            out["two_level"] = two_level(A, subs, args, np, mu, reps)
        elif tl:
            out["two_level"] = tl
        if want_cpu:
            out["cpu_baseline"] = cpu_baseline(A, subs, d, args, np)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def two_level(A, subs, args, np, mu, reps):
    """configs[2] flavour on the same operator: deflated two-level apply with nu deflation vectors per subdomain.  GenEO's
    eigensolver is a later row of SURVEY section 8(f); for kernel timing the vectors are the 20 monomials of degree <= 3 in
    the local coordinates, the coarse operator E = Z^T A Z is assembled and inverted as in the reference."""
    nu = args.geneo_nu
    expo = [(a, b, c) for deg in range(8) for a in range(deg + 1) for b in range(deg + 1 - a) for c in [deg - a - b]][:nu]
    tg = time.time()
    lam_max = None
    for s, sd in enumerate(subs):
        if args.geneo:
            A.set_option("geneo_nu", nu)
            lam = A.solve_gevp(s, sd["n"], sd.get("ia_neumann", sd["ia"]), sd.get("ja_neumann", sd["ja"]), sd["a_neumann"], sd["sym"])
            lam_max = max(lam_max or 0.0, float(lam[-1]))
            continue
        i0, i1, j0, j1, k0, k1 = sd["box"]
        z, y, x = np.meshgrid(np.linspace(-1, 1, k1 - k0), np.linspace(-1, 1, j1 - j0), np.linspace(-1, 1, i1 - i0), indexing="ij")
        if sd.get("block", 1) == 3:
            # stand-in for elasticity: monomials on each displacement component (the first 12 span the rigid-body modes)
            P = np.stack([(x ** a * y ** b * z ** c).ravel() for a, b, c in expo[:(nu + 2) // 3]], axis=1)
            Z = np.zeros((sd["n"], 3 * P.shape[1]))
            for comp in range(3):
                Z[comp::3, comp::3] = P
            Z = Z[:, :nu]
        else:
            Z = np.stack([(x ** a * y ** b * z ** c).ravel() for a, b, c in expo], axis=1)
        A.set_vectors(s, Z)
    tg = time.time() - tg
    t0 = time.time()
    A.build_coarse_operator()
    t_coarse = time.time() - t0
    A.set_option("schwarz_coarse_correction", 0)  # deflated
    t_defl = A.time("deflation", mu=mu, warmup=2, reps=reps)
    t_apply = A.time("apply", mu=mu, warmup=2, reps=reps)
    import torch
    f = torch.from_numpy(np.concatenate([s["f"] for s in subs])).to(torch.device("cuda", torch.cuda.current_device()))
    xs = torch.zeros_like(f)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    it2 = A.solve_device(f.data_ptr(), xs.data_ptr(), 1)
    torch.cuda.synchronize()
    t_gm = time.perf_counter() - t1
    A.set_option("schwarz_coarse_correction", -1)
    n = A.stats()["n"]
    bytes_panel = 2.0 * n * nu * 8.0 + 3.0 * n * mu * 8.0   # Z read twice + D r read, Z y written, ...
    return {"geneo_nu": nu, "coarse_dim": int(A.stats()["coarse_dim"]), "deflation_ms": t_defl * 1e3, "apply_ms": t_apply * 1e3,
            "applies_per_sec": 1.0 / t_apply, "deflation_panel_GBps": bytes_panel / t_defl / 1e9, "deflation_flops": 4.0 * n * nu * mu,
            "coarse_setup_seconds": round(t_coarse, 2),
            "kernel": "k_zt_stream + k_z_stream (GEMV-shaped: streaming VALU)" if mu <= 2 else "k_zt_mfma + k_z_mfma (v_mfma_f64_16x16x4_f64)",
            "coarse_space": ("GenEO (solveGEVP), largest kept eigenvalue %.3f, %.1f s" % (lam_max, tg)) if args.geneo else "monomials of degree <= 3 (stand-in)",
            "gmres": {"iterations": it2, "seconds": t_gm, "iters_per_sec": it2 / t_gm}}


def cpu_baseline(A, subs, d, args, np):
    """the oracle's substitution (plain C, oracle/sptrsv_oracle.c) on the SAME factors, one host thread per subdomain
    (the reference runs one MPI rank per subdomain with a sequential local solve), plus the numpy halo sum"""
    from oracle import sptrsv_oracle
    from oracle.ras_oracle import Oracle
    nsub = len(subs)
    threads = min(nsub, os.cpu_count() or 1)
    factors = [sptrsv_oracle.PlainFactor(A.subdomain(s)) for s in range(nsub)]
    orc = Oracle(subs)
    orc.d = d
    f = [np.ones(s["n"]) for s in subs]
    sptrsv_oracle.time_batch(factors, f, reps=1, threads=threads)  # warm-up
    reps = 0
    tsolve = tex = 0.0
    t_begin = time.perf_counter()
    while time.perf_counter() - t_begin < 12.0 and reps < 20:
        sec, xs = sptrsv_oracle.time_batch(factors, f, reps=1, threads=threads)
        t1 = time.perf_counter()
        orc.exchange(xs)
        tex += time.perf_counter() - t1
        tsolve += sec
        reps += 1
    per_apply = (tsolve + tex) / reps
    return {"value": 1.0 / per_apply, "unit": "applies/s", "cores": threads, "kind": "port",
            "sample": f"{reps} full applies of the same {nsub}-subdomain operator (all {nsub} local substitutions on {threads} threads, "
                      f"one per subdomain, + numpy halo sum); substitution {tsolve / reps * 1e3:.1f} ms, halo {tex / reps * 1e3:.1f} ms per apply",
            "seconds_per_apply": per_apply}


if __name__ == "__main__":
    main()
