#!/usr/bin/env python3
"""developer aid: numfact time of one subdomain (N^3 7-point Laplacian) for the three kinds, upper levels on the device or on the host.
usage: time_numfact.py [N=65] [kinds=chol,ldlt,lu] [where=device,host]"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hpddm_amd import hpddm  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65
I = sp.identity(N)
T = sp.diags([-1, 2, -1], [-1, 0, 1], shape=(N, N))
A = (sp.kron(sp.kron(T, I), I) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(I, I), T)).tocsr()
n = A.shape[0]
rng = np.random.default_rng(0)
cases = {"chol": (sp.tril(A).tocsr(), True, True), "ldlt": (sp.tril(A - 0.05 * sp.identity(n)).tocsr(), True, False),
         "lu": ((A + 0.2 * sp.triu(A, 1) + sp.diags(rng.random(n))).tocsr(), False, False)}
kinds = sys.argv[2].split(",") if len(sys.argv) > 2 else list(cases)
for where in (sys.argv[3].split(",") if len(sys.argv) > 3 else ("device", "host")):
    if where == "host":
        os.environ["HPDDM_HIP_HOST_FACTOR"] = "1"
    for name in kinds:
        M, sym, spd = cases[name]
        M.sort_indices()
        S = hpddm.Subdomain()
        S.numfact(n, M.indptr, M.indices, M.data, sym=sym, spd=spd)   # analysis + first factorisation
        t0 = time.time()
        S.numfact(n, M.indptr, M.indices, M.data, sym=sym, spd=spd)   # refactorisation: numerical phase only
        t = time.time() - t0
        info = S.info()
        b = rng.random(n)
        x = S.solve(b)
        full = M + sp.tril(M, -1).T if sym else M
        print(f"{where:6s} {name:5s} N={N}: numfact {t:6.2f} s (numeric {info['t_numeric']:.2f} s), kind {info['kind']}, residual {np.abs(full @ x - b).max():.2e}", flush=True)
        S.destroy()
