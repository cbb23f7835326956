#!/usr/bin/env python3
"""developer aid: factorise the bench operator ONCE, then rebuild the SpTRSV level schedule under several settings of the
plan-builder knobs (HPDDM_HIP_* environment variables) and time the batched sweep pair for each.
usage: sweep_plan.py [--grid 256 | --helmholtz 64,64,128] [--levels] [--mu 1] "K1=v K2=v" "K1=w" ...      ("" = defaults)"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--mu", default="1", help="right-hand sides; a comma-separated list times each setting for every entry")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--helmholtz", default=None, help="nx,ny,nz: the complex shifted Laplacian of bench.py --problem helmholtz instead")
    ap.add_argument("--levels", action="store_true", help="print the per-launch table for every setting")
    ap.add_argument("--options", default="", help="appended to the options of the operator, e.g. '-hpddm_hip_numfact_threads 1'")
    ap.add_argument("cfgs", nargs="*", default=[""])
    args = ap.parse_args()
    from hpddm_amd import hpddm
    from hpddm_amd.generate import generate3d
    hpddm.require_device()
    t0 = time.time()
    if args.helmholtz:
        from hpddm_amd.generate import generate_helmholtz3d
        subs = generate_helmholtz3d(tuple(int(v) for v in args.helmholtz.split(",")), 8, grid=(2, 2, 2))   # configs[4]'s share: ORAS on the impedance matrices
        A, d = hpddm.schwarz_from_subdomains(subs, options="-hpddm_schwarz_method oras " + args.options, multiplicity=False)
        for s_, sd in enumerate(subs):
            A.set_optimized_matrix(s_, sd["n"], sd["ia"], sd["ja"], sd["a_opt"], False)
    else:
        subs = generate3d(args.grid, 8, overlap=1, sym=True, rhs="smooth")
        A, d = hpddm.schwarz_from_subdomains(subs, options="-hpddm_operator_spd " + args.options)
    A.call_numfact()
    st = A.stats()
    print(f"setup {time.time() - t0:.1f} s, nnz(L) {st['nnz_L']:.4g}, levels {int(st['levels'])}", flush=True)
    sk = 16.0 if args.helmholtz else 8.0
    mus = [int(v) for v in args.mu.split(",")]
    touched = set()
    for cfg in args.cfgs:
        for k in touched:
            os.environ.pop(k, None)
        for kv in cfg.split():
            k, v = kv.split("=")
            os.environ[k] = v
            touched.add(k)
        A.rebuild_plan()
        for mu in mus:
            bytes_alg = 2.0 * st["nnz_L"] * sk + 4.0 * st["n"] * mu * sk
            t = A.time("solve", mu=mu, warmup=2, reps=args.reps)
            print(f"== [{cfg}]  mu {mu}  sptrsv {t * 1e3:.3f} ms  frac {bytes_alg / t / 8e12:.4f}  launches {int(A.stats()['launches'])}", flush=True)
            if args.levels and os.environ.get("HPDDM_HIP_STREAMS") == "1":
                tot = {"fwd": [0.0, 0.0], "bwd": [0.0, 0.0]}
                for kind, lev, us, nbytes in A.level_times(mu=mu, reps=3):
                    extra = f"  {nbytes / 1e6:9.1f} MB {nbytes / us / 1e3:8.1f} GB/s" if nbytes else ""
                    print(f"   {kind:8s} level {lev:2d} {us:9.1f} us{extra}")
                    k2 = kind.replace("_chain", "")
                    if k2 in tot:
                        tot[k2][0] += us
                        tot[k2][1] += nbytes
                for k, (us, nb) in tot.items():
                    print(f"   {k} total {us:9.1f} us {nb / 1e6:9.1f} MB {nb / us / 1e3:8.1f} GB/s")


if __name__ == "__main__":
    main()
