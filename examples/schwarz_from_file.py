#!/usr/bin/env python3
"""Counterpart of the reference's examples/schwarzFromFile.cpp + generateFromFile.cpp: a global matrix read from a file in the
reference's text format (hpddm_amd/matrix_io.py), decomposed algebraically into overlapping subdomains (hpddm_amd/decompose.py -- the
reference partitions with METIS, here strips of a reverse Cuthill-McKee ordering), solved with the Schwarz-preconditioned Krylov method
of the -hpddm_* options, true residual of the global system printed like the reference does.

    python examples/schwarz_from_file.py -matrix_filename=tests/golden/dump/out_0_4.txt --subdomains 3 -overlap 2 -hpddm_verbosity=1
    python examples/schwarz_from_file.py -matrix_filename=A.txt -rhs_filename=b.txt --subdomains 8 -hpddm_schwarz_method asm -hpddm_krylov_method cg
"""
import os
import sys

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hpddm_amd import hpddm  # noqa: E402
from hpddm_amd.decompose import decompose, gather  # noqa: E402
from hpddm_amd.matrix_io import read_matrix  # noqa: E402


def parse(argv):
    app, lib, i = {"matrix_filename": None, "rhs_filename": None, "subdomains": "4", "overlap": "1"}, [], 0
    while i < len(argv):
        t = argv[i]
        key, val = (t.lstrip("-").split("=", 1) + [None])[:2] if "=" in t else (t.lstrip("-"), None)
        if t.startswith("-hpddm_"):
            lib.append(t)
            if val is None and i + 1 < len(argv) and not argv[i + 1].startswith("-"):
                i += 1
                lib.append(argv[i])
        elif key in app:
            if val is None:
                i += 1
                val = argv[i]
            app[key] = val
        else:
            raise SystemExit(f"unknown option {t}")
        i += 1
    if not app["matrix_filename"]:
        raise SystemExit("-matrix_filename=<file> is required")
    return app, " ".join(lib)


def read_rhs(path, n):
    """one value per line; an optional first line holding n is skipped (generateFromFile.cpp:143-158)"""
    vals = [ln.split()[0] for ln in open(path) if ln.strip()]
    if len(vals) == n + 1 and float(vals[0]) == n:
        vals = vals[1:]
    if len(vals) != n:
        raise SystemExit(f"{path}: expected {n} values, found {len(vals)}")
    return np.array(vals, dtype=np.float64)


def main(argv):
    app, lib = parse(argv)
    mat = read_matrix(app["matrix_filename"])
    n = mat["n"]
    A = sp.csr_matrix((mat["a"], mat["ja"], mat["ia"]), shape=(n, mat["m"]))
    if mat["sym"]:   # lower triangle stored: expand, the decomposition works on general storage
        A = (A + sp.tril(A, -1).T).tocsr()
    b = read_rhs(app["rhs_filename"], n) if app["rhs_filename"] else np.random.default_rng(0).uniform(0.0, 10.0, n)
    parts = int(app["subdomains"])
    if parts == 1:   # a single subdomain is a direct solve, like in the reference's drivers
        S = hpddm.Subdomain()
        S.numfact(n, A.indptr, A.indices, A.data, sym=False)
        x = S.solve(b)
    else:
        subs = decompose(A, parts, int(app["overlap"]), rhs=b)
        op, _ = hpddm.schwarz_from_subdomains(subs, options=lib)
        op.call_numfact()
        it, sol = op.solve([sd["f"] for sd in subs])
        x = gather(subs, sol, n)
        op.destroy()
    nrmb, nrmAx = np.linalg.norm(b), np.linalg.norm(A @ x - b)
    print(" --- residual = {:e} / {:e}".format(nrmAx, nrmb))
    return 1 if nrmAx / nrmb > 1.0e-4 else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
