"""The small dense eigen-solver behind GCRO-DR's harmonic Ritz problems (hpddm_amd/csrc/dense_eig.cpp, host code of the
library) against numpy: residual of every eigenpair and distance of every eigenvalue to LAPACK's.  CPU only."""
import ctypes

import numpy as np
import pytest

from hpddm_amd import _lib


def _eig(A):
    L = _lib.load()
    n = A.shape[0]
    A = np.ascontiguousarray(A, dtype=np.float64)
    wr, wi, V = np.zeros(n), np.zeros(n), np.zeros((n, n))
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    assert L.HpddmHipDenseEig(n, p(A), p(wr), p(wi), p(V)) == 0
    lam = wr + 1j * wi
    Vc = np.zeros((n, n), dtype=complex)
    j = 0
    while j < n:
        if wi[j] == 0:
            Vc[:, j] = V[:, j]
            j += 1
        else:   # complex pair: columns j, j+1 = real and imaginary parts of the vector of the first one
            assert wi[j] > 0 > wi[j + 1]
            Vc[:, j], Vc[:, j + 1] = V[:, j] + 1j * V[:, j + 1], V[:, j] - 1j * V[:, j + 1]
            j += 2
    return lam, Vc


@pytest.mark.parametrize("n", [1, 2, 3, 5, 8, 10, 13, 20, 40])
def test_eigenpairs_of_random_matrices(n):
    rng = np.random.default_rng(100 + n)
    for trial in range(12):
        A = rng.standard_normal((n, n))
        if trial % 3 == 1:
            A = np.triu(A, -1)          # Hessenberg, like the first-cycle problem
        elif trial % 3 == 2:
            A = A + A.T                 # real spectrum
        lam, V = _eig(A)
        assert np.abs(A @ V - V * lam).max() <= 1e-10 * max(1.0, np.abs(A).max()) * max(1.0, np.abs(V).max())
        ref = np.linalg.eigvals(A)
        assert max(np.min(np.abs(ref - l)) for l in lam) <= 1e-9 * max(1.0, np.abs(ref).max())


def test_defective_and_zero_matrices():
    lam, V = _eig(np.zeros((4, 4)))
    assert np.all(lam == 0)
    J = np.diag(np.ones(3), 1) + 2.0 * np.eye(4)   # one Jordan block
    lam, V = _eig(J)
    assert np.allclose(lam, 2.0, atol=1e-3)


def test_host_helpers_of_the_recycling_methods():
    """Householder QR, triangular inverse, the six -hpddm_recycle_target orders and the selection of Ritz vectors (whole and cut
    complex pairs) used by GCRO-DR / Block GCRO-DR, and the real-equivalent embedding of complex matrices and deflation vectors against
    complex arithmetic: the library's own host self-test"""
    assert _lib.load().HpddmHipHostSelfTest() == 0


def _eig_z(A):
    L = _lib.load()
    n = A.shape[0]
    A = np.ascontiguousarray(A, dtype=np.complex128)
    w, V = np.zeros(n, dtype=np.complex128), np.zeros((n, n), dtype=np.complex128)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    assert L.HpddmHipDenseEigZ(n, p(A), p(w), p(V)) == 0
    return w, V


@pytest.mark.parametrize("n", [1, 2, 3, 5, 8, 13, 24, 60, 150])
def test_complex_eigenpairs(n):
    """the Rayleigh-Ritz problems of the complex GenEO eigensolver: general complex, complex symmetric, block Hessenberg, Hermitian"""
    rng = np.random.default_rng(300 + n)
    for trial in range(8 if n < 100 else 2):
        A = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        if trial % 4 == 1:
            A = A + A.T                 # complex symmetric (the pencils of a Helmholtz subdomain)
        elif trial % 4 == 2:
            A = np.triu(A, -4)          # block upper Hessenberg (block Arnoldi with 4 columns)
        elif trial % 4 == 3:
            A = A + A.conj().T          # Hermitian: real spectrum
        lam, V = _eig_z(A)
        assert np.abs(A @ V - V * lam).max() <= 1e-9 * max(1.0, np.abs(A).max()) * n
        assert np.allclose(np.linalg.norm(V, axis=0), 1.0)
        ref = np.linalg.eigvals(A)
        assert max(np.min(np.abs(ref - l)) for l in lam) <= 1e-8 * max(1.0, np.abs(ref).max())
        assert max(np.min(np.abs(lam - r)) for r in ref) <= 1e-8 * max(1.0, np.abs(ref).max())


def test_complex_degenerate_matrices():
    lam, V = _eig_z(np.zeros((3, 3)))
    assert np.all(lam == 0) and np.allclose(V, np.eye(3))
    lam, V = _eig_z(np.diag([1 + 1j, 1 + 1j, 2.0]))      # a double eigenvalue with two vectors
    assert np.allclose(sorted(lam, key=lambda x: x.real), [1 + 1j, 1 + 1j, 2.0])
    lam, V = _eig_z(np.diag(np.ones(3), 1) + (2.0 - 1j) * np.eye(4))   # one Jordan block
    assert np.allclose(lam, 2.0 - 1j, atol=1e-3)
