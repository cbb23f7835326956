"""oracle/sptrsv_oracle.c (CPU substitution on the exported plain factor) pinned against scipy's SuperLU."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spl

from hpddm_amd import hpddm
from oracle import sptrsv_oracle


def _poisson3d(N):
    I = sp.identity(N)
    T = sp.diags([-1, 2, -1], [-1, 0, 1], shape=(N, N))
    return (sp.kron(sp.kron(T, I), I) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(I, I), T)).tocsr()


@pytest.mark.parametrize("kind", ["chol", "ldlt", "lu"])
def test_cpu_substitution_matches_superlu(kind):
    A = _poisson3d(10)
    n = A.shape[0]
    rng = np.random.default_rng(1)
    if kind == "chol":
        Ain, sym, spd = sp.tril(A).tocsr(), True, True
    elif kind == "ldlt":
        A = (A - 1.7 * sp.identity(n)).tocsr()
        Ain, sym, spd = sp.tril(A).tocsr(), True, False
    else:
        A = (A + sp.diags(rng.random(n)) + 0.3 * sp.triu(A, 1)).tocsr()
        Ain, sym, spd = A, False, False
    Ain.sort_indices()
    S = hpddm.Subdomain(host_only=1, keep_plain=1)
    S.numfact(n, Ain.indptr, Ain.indices, Ain.data, sym=sym, spd=spd)
    pf = sptrsv_oracle.PlainFactor(S)
    b = np.asfortranarray(rng.random((n, 3)))
    x = pf.solve(b)
    ref = spl.splu(sp.csc_matrix(A)).solve(b)
    assert np.abs(x - ref).max() <= 1e-11 * np.abs(ref).max()
    sec, xs = sptrsv_oracle.time_batch([pf, pf], [b, b], reps=2, threads=2)
    assert np.abs(xs[0] - ref).max() <= 1e-11 * np.abs(ref).max() and sec > 0
    S.destroy()
