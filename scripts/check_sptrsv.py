#!/usr/bin/env python3
"""developer aid (GPU): the local solve of one subdomain against SciPy's SuperLU on a handful of matrices -- real Cholesky / L D L^T /
LU, complex L D L^T / LU, 1 ... 17 right-hand sides (the register-blocked VALU sweeps and the 16-column engine), with the leaves
condensed and not; prints every case and exits non-zero if one misses 1e-9.  A faster, more talkative first stop than the test suite."""
import os
import sys
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def poisson3d(N):
    I = sp.identity(N)
    T = sp.diags([-1, 2, -1], [-1, 0, 1], shape=(N, N))
    return (sp.kron(sp.kron(T, I), I) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(I, I), T)).tocsr()


def main():
    from hpddm_amd import hpddm
    hpddm.require_device()
    rng = np.random.default_rng(5)
    bad = 0
    sizes = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "9,24").split(",")]
    for N in sizes:
        K = poisson3d(N)
        n = K.shape[0]
        cases = [("chol", sp.tril(K).tocsr(), True, True, K),
                 ("ldlt", sp.tril(K - 0.31 * sp.identity(n)).tocsr(), True, False, (K - 0.31 * sp.identity(n)).tocsr()),
                 ("lu", (K + sp.diags(rng.random(n)) + 0.3 * sp.triu(K, 1)).tocsr(), False, False, None),
                 ("z-ldlt", sp.tril(K - (0.31 - 0.2j) * sp.identity(n)).tocsr().astype(np.complex128), True, False, (K - (0.31 - 0.2j) * sp.identity(n)).tocsr()),
                 ("z-lu", (K + 0.2 * sp.triu(K, 1) + 0.3j * sp.diags(rng.random(n))).tocsr().astype(np.complex128), False, False, None)]
        for name, Ain, sym, spd, full in cases:
            full = Ain if full is None else full
            Ain.sort_indices()
            lu = spla.splu(sp.csc_matrix(full))
            cplx = np.iscomplexobj(Ain.data)
            for condense in (1, 0):
                os.environ["HPDDM_HIP_CONDENSE"] = str(condense)
                S = hpddm.Subdomain()
                S.numfact(n, Ain.indptr, Ain.indices, Ain.data, sym=sym, spd=spd)
                nleaf = int((S.export("lb_off") >= 0).sum())
                for mu in (1, 2, 3, 5, 8, 9, 16, 17):
                    b = rng.random((mu, n)) + (1j * rng.random((mu, n)) if cplx else 0)
                    x = np.asarray(S.solve(np.asfortranarray(b.T))).T
                    ref = np.stack([lu.solve(b[k]) for k in range(mu)])
                    err = np.abs(x - ref).max() / np.abs(ref).max()
                    ok = err < 1e-9
                    bad += not ok
                    print(f"{'ok  ' if ok else 'FAIL'} N={N:3d} {name:7s} condensed leaves {nleaf:6d}  mu={mu:2d}  rel. error {err:.2e}", flush=True)
                S.destroy()
    os.environ.pop("HPDDM_HIP_CONDENSE", None)
    print("failures:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
