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
    # the all-core variant (level-scheduled over the assembly tree, atomic extend-adds): same solution
    sec, xl = sptrsv_oracle.time_batch_levels([pf, pf], [b[:, 0].copy(), b[:, 1].copy()], reps=2, threads=4)
    assert np.abs(xl[0] - ref[:, 0]).max() <= 1e-11 * np.abs(ref).max() and np.abs(xl[1] - ref[:, 1]).max() <= 1e-11 * np.abs(ref).max() and sec > 0
    S.destroy()


def test_level_parallel_substitution_large_supernodes():
    """a size where the team kernels (supernodes of more than 2^18 entries) of the all-core baseline are exercised"""
    A = _poisson3d(40)
    n = A.shape[0]
    Ain = sp.tril(A).tocsr()
    Ain.sort_indices()
    S = hpddm.Subdomain(host_only=1, keep_plain=1)
    S.numfact(n, Ain.indptr, Ain.indices, Ain.data, sym=True, spd=True)
    pf = sptrsv_oracle.PlainFactor(S)
    w = np.diff(pf.arr["blk_ptr"])
    h = w + np.diff(pf.arr["row_ptr"])
    assert (w * (w + 1) // 2 + (h - w) * w).max() >= 1 << 18
    b = np.random.default_rng(3).random(n)
    ref = pf.solve(b)
    assert np.abs(A @ ref - b).max() < 1e-9
    for threads in (1, 3, 8):
        _, xl = sptrsv_oracle.time_batch_levels([pf], [b], reps=1, threads=threads)
        assert np.abs(xl[0] - ref).max() <= 1e-12 * np.abs(ref).max()
    # teams of threads per subdomain (nested OpenMP on the row loops of the large supernodes): same solution
    for team in (1, 2, 3):
        _, xt = sptrsv_oracle.time_batch_teams([pf, pf], [b, b], reps=1, team=team)
        assert np.abs(xt[0] - ref).max() <= 1e-12 * np.abs(ref).max() and np.abs(xt[1] - ref).max() <= 1e-12 * np.abs(ref).max()
    S.destroy()


@pytest.mark.parametrize("kind", ["ldlt", "lu"])
def test_complex_cpu_substitution_matches_superlu(kind):
    """the same port for K = std::complex<double> (solve_one_z: complex symmetric L D L^T with plain transposes, complex LU) -- the CPU
    leg of bench.py --problem helmholtz (configs[4]'s share)"""
    K = _poisson3d(9)
    n = K.shape[0]
    rng = np.random.default_rng(4)
    if kind == "ldlt":
        A = (K - (1.3 - 0.4j) * sp.identity(n)).tocsr()
        Ain, sym = sp.tril(A).tocsr(), True
    else:
        A = (K + 0.2 * sp.triu(K, 1) + 0.3j * sp.diags(rng.random(n))).tocsr()
        Ain, sym = A, False
    Ain.sort_indices()
    S = hpddm.Subdomain(host_only=1, keep_plain=1)
    S.numfact(n, Ain.indptr, Ain.indices, Ain.data.astype(np.complex128), sym=sym)
    pf = sptrsv_oracle.PlainFactor(S)
    assert pf.complex and pf.kind == (1 if kind == "ldlt" else 2)
    b = np.asfortranarray(rng.random((n, 3)) + 1j * rng.random((n, 3)))
    sec, xs = sptrsv_oracle.time_batch_z([pf, pf], [b, b], reps=2, threads=2)
    ref = spl.splu(sp.csc_matrix(A)).solve(b)
    assert np.abs(xs[0] - ref).max() <= 1e-11 * np.abs(ref).max() and np.abs(xs[1] - ref).max() <= 1e-11 * np.abs(ref).max() and sec > 0
    # the block of right-hand sides goes through solve_block_z (every factor entry read once per block), a single one through
    # solve_one_z: per column the same operations in the same order (the compiler contracts the multiply-adds of the two loops differently)
    for c in range(3):
        _, x1 = sptrsv_oracle.time_batch_z([pf], [np.asfortranarray(b[:, c:c + 1])], reps=1, threads=1)
        assert np.abs(x1[0][:, 0] - xs[0][:, c]).max() <= 1e-13 * np.abs(ref).max()
    S.destroy()
