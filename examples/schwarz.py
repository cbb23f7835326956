#!/usr/bin/env python3
"""Counterpart of the reference's examples/schwarz.py (itself the Python twin of examples/schwarz.cpp:81-196) on this library:
the 2-D Poisson problem of examples/generate.py split into overlapping subdomains, one- or two-level Schwarz preconditioner,
Krylov solve, residual check.  The reference runs one subdomain per MPI rank (`mpirun -np 4 python examples/schwarz.py ...`); here
one process holds them all on one GPU, so the count is an option:

    python examples/schwarz.py --subdomains 4 -Nx 200 -Ny 200 -hpddm_verbosity=1            # GMRES, 45 iterations
    python examples/schwarz.py --subdomains 4 -Nx 40 -Ny 40 -hpddm_schwarz_coarse_correction deflated -hpddm_geneo_nu=0
    python examples/schwarz.py --subdomains 4 -Nx 40 -Ny 40 -generate_random_rhs 4 -hpddm_krylov_method bgmres

Application options as in the reference (-Nx -Ny -overlap -generate_random_rhs -symmetric_csr), every -hpddm_* option is handed to
the library unchanged.  Exit status like the reference: 1 if the solve took more than 45 iterations (60 for bfbcg) or a relative
residual exceeds 1e-2.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hpddm_amd import hpddm  # noqa: E402
from hpddm_amd.generate import generate2d  # noqa: E402

APP_INT = {"Nx": 100, "Ny": 100, "overlap": 1, "generate_random_rhs": 0, "symmetric_csr": 0}


def parse(argv):
    """-> (application options, number of subdomains, the -hpddm_* part of the command line)"""
    app, size, lib, i = dict(APP_INT), 4, [], 0
    while i < len(argv):
        t = argv[i]
        key, val = (t.lstrip("-").split("=", 1) + [None])[:2] if "=" in t else (t.lstrip("-"), None)
        if t.startswith("-hpddm_"):
            lib.append(t)
            if val is None and i + 1 < len(argv) and not argv[i + 1].startswith("-"):
                i += 1
                lib.append(argv[i])
        elif key in APP_INT or key == "subdomains":
            if val is None:
                i += 1
                val = argv[i]
            if key == "subdomains":
                size = int(val)
            else:
                app[key] = int(val)
        else:
            raise SystemExit(f"unknown option {t}")
        i += 1
    return app, size, " ".join(lib)


def main(argv):
    app, size, lib = parse(argv)
    subs = generate2d(app["Nx"], app["Ny"], size, overlap=app["overlap"], sym=bool(app["symmetric_csr"]))
    mu = app["generate_random_rhs"]
    if size == 1:   # examples/schwarz.py:104-131: a single subdomain is a direct solve
        sd = subs[0]
        S = hpddm.Subdomain()
        S.numfact(sd["n"], sd["ia"], sd["ja"], sd["a"], sym=sd["sym"])
        sol = S.solve(sd["f"])
        from hpddm_amd.matrix_io import csrmv
        nrmb, nrmAx = np.linalg.norm(sd["f"]), np.linalg.norm(csrmv(sd, sol) - sd["f"])
        print(" --- residual = {:e} / {:e}".format(nrmAx, nrmb))
        return 1 if nrmAx / nrmb > 1.0e-6 else 0
    A, d = hpddm.schwarz_from_subdomains(subs, options=lib)
    if mu:
        rng = np.random.default_rng(0)
        f = A.exchange([rng.random((sd["n"], mu)) for sd in subs])   # random right-hand sides made consistent on the overlap (schwarzExchange)
    else:
        f, mu = [sd["f"] for sd in subs], 1
    if "schwarz_coarse_correction" in lib:
        if "geneo_nu" not in lib or A.get_option("geneo_nu") > 0:   # the reference's default is 20 GenEO vectors
            raise SystemExit("the 2-D generator carries no Neumann matrices: pass -hpddm_geneo_nu=0 (constant deflation vector, like the "
                             "reference built without an eigensolver), or see bench.py --geneo for GenEO")
        for s, sd in enumerate(subs):
            A.set_vectors(s, np.ones((sd["n"], 1)))
        A.build_coarse_operator()
    A.call_numfact()
    it, sol = A.solve(f)
    storage = A.compute_residual(sol, f)
    for nu in range(mu):
        print(("                " if nu else " --- residual = ") + "{:e} / {:e}".format(storage[1 + 2 * nu], storage[2 * nu]) + (" (rhs #{:d})".format(nu + 1) if mu > 1 else ""))
    status = 1 if it > (60 if int(A.get_option("krylov_method")) == 6 else 45) else int(any(storage[1 + 2 * nu] / storage[2 * nu] > 1.0e-2 for nu in range(mu)))
    A.destroy()
    return status


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
