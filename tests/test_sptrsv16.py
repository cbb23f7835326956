"""The 16-column MFMA engine of the local solve (hpddm_amd/csrc/sptrsv16.hip: 8 complex or 16 real right-hand sides per sweep,
interleaved vectors) against SciPy's SuperLU -- Solver<K>::solve with n right-hand sides (include/HPDDM_MUMPS.hpp:304-317) for the
three kinds of factorisation, real and complex scalars, block counts that mix the engine with the register-blocked VALU sweeps
(16 + 3, 2 x 16 + 3, 8 + 1 complex ...), narrow and wide panels, split-row tiles, in place and out of place."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spl

from hpddm_amd import hpddm

pytestmark = pytest.mark.gpu


def _laplace3d(nx, ny=None, nz=None):
    ny, nz = ny or nx, nz or nx
    T = lambda n: sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(n, n))
    I = sp.identity
    return (sp.kron(sp.kron(T(nz), I(ny)), I(nx)) + sp.kron(sp.kron(I(nz), T(ny)), I(nx)) + sp.kron(sp.kron(I(nz), I(ny)), T(nx))).tocsr()


def _solve_and_check(A, sym, spd, mus, cplx=False, tol=1e-9):
    n = A.shape[0]
    M = (sp.tril(A, format="csr") if sym else A.tocsr())
    M.sort_indices()
    S = hpddm.Subdomain()
    S.numfact(n, M.indptr, M.indices, M.data.astype(np.complex128) if cplx else M.data, sym=sym, spd=spd)
    lu = spl.splu(A.tocsc().astype(np.complex128 if cplx else np.float64))
    rng = np.random.default_rng(8)
    for mu in mus:
        b = rng.standard_normal((n, mu)) + (1j * rng.standard_normal((n, mu)) if cplx else 0.0)
        b = np.asfortranarray(b)
        x = S.solve(b)
        ref = lu.solve(np.asarray(b))
        err = np.abs(x - ref).max(axis=0) / np.abs(ref).max(axis=0)
        assert err.max() <= tol, (mu, err)
        # every column on its own gives the same answer as inside the block of 16 (the engine and the VALU sweeps agree)
        x1 = S.solve(np.asfortranarray(b[:, mu - 1]))
        assert np.abs(x1 - x[:, mu - 1]).max() <= 1e-11 * np.abs(x1).max()
    kind = S.info()["kind"]
    S.destroy()
    return kind


def test_sixteen_real_right_hand_sides_three_kinds():
    n = 14
    K = _laplace3d(n)
    N = n ** 3
    rng = np.random.default_rng(2)
    assert _solve_and_check(K, True, True, (16, 19, 35, 10, 12, 26)) == 0                                   # Cholesky; 10, 12, 26 = 16 + 10: a last block of 10 to 15 columns takes the engine too, zero columns beside it
    assert _solve_and_check((K - 0.35 * sp.identity(N)).tocsr(), True, False, (16, 17)) == 1                   # LDL^T (indefinite shift)
    assert _solve_and_check((K + 0.2 * sp.triu(K, 1) + sp.diags(rng.random(N))).tocsr(), False, False, (16, 33)) == 2   # LU


def test_eight_complex_right_hand_sides_ldlt_and_lu():
    n = 12
    K = _laplace3d(n) * float(n * n)
    N = n ** 3
    k2 = (2.5 * np.pi) ** 2
    A = (K - k2 * sp.identity(N) + 1j * 0.8 * k2 * sp.identity(N)).tocsr()
    assert _solve_and_check(A, True, False, (8, 9, 16, 21, 5, 7), cplx=True) == 1      # complex symmetric: L D L^T (21 = 8 + 8 + 5; 5 to 7: one padded block of the engine)
    G = sp.random(N, N, density=2e-3, random_state=7, format="csr")
    B = (K + 0.2 * sp.triu(K, 1) + 1j * 0.1 * float(n * n) * G).tocsr()
    assert _solve_and_check(B, False, False, (8, 13), cplx=True) == 2            # general complex: LU (separate backward panels)


def test_wide_panels_and_split_rows_sixteen_columns():
    """separators of 30 x 30 and 30 x 15 cells: panels wider than 128 columns (block tiles), upper backward levels split in parts"""
    K = _laplace3d(30, 30, 31)
    assert _solve_and_check(K, True, True, (16,), tol=1e-8) == 0
    n = 20
    Kc = _laplace3d(n, n, 2 * n) * float(n * n)
    A = (Kc - 0.03 * sp.diags(Kc.diagonal()) + 0.03j * sp.diags(Kc.diagonal())).tocsr()   # the Helmholtz-like operator of bench.py
    assert _solve_and_check(A, False, False, (8,), cplx=True, tol=1e-8) in (1, 2)


def test_in_place_and_batched_subdomains():
    """the batched plan of a Schwarz operator (8 subdomains, several groups on several streams) on 16 real right-hand sides, and the
    in-place call of the Solver concept (x aliases b)"""
    from hpddm_amd.generate import generate3d
    subs = generate3d(16, 8, overlap=1, sym=True, rhs="smooth")
    A, d = hpddm.schwarz_from_subdomains(subs, options="-hpddm_operator_spd")
    A.call_numfact()
    rng = np.random.default_rng(3)
    f = [np.asfortranarray(rng.standard_normal((s["n"], 16))) for s in subs]
    x = A.local_solve(f)
    for s, sd in enumerate(subs):
        M = sp.csr_matrix((sd["a"], sd["ja"], sd["ia"]), shape=(sd["n"], sd["n"]))
        full = M + sp.tril(M, -1).T
        ref = spl.splu(full.tocsc()).solve(f[s])
        assert np.abs(x[s] - ref).max() <= 1e-10 * np.abs(ref).max()
    A.destroy()


@pytest.mark.parametrize("mu", [1, 16])
def test_split_row_hand_over_is_bitwise_reproducible(mu):
    """the upper backward levels are split over several workgroups that publish partial sums (write-through stores) and meet at an
    agent-scope arrival counter; the last arriver adds the parts in part order.  The same solve repeated 150 times, with other
    work in flight on the same GPU in between, must return the same bits every time (both the VALU tiles, mu = 1, and the
    16-column engine) -- a stale or torn partial sum would show up as a difference"""
    from hpddm_amd.generate import generate3d
    subs = generate3d(40, 8, overlap=1, sym=True, rhs="smooth")     # 21^3 per subdomain: wide separators, split parts on the top levels
    A, d = hpddm.schwarz_from_subdomains(subs, options="-hpddm_operator_spd")
    A.call_numfact()
    rng = np.random.default_rng(12)
    f = [np.asfortranarray(rng.standard_normal((s["n"], mu))) for s in subs]
    ref = A.local_solve(f)
    g = [np.asfortranarray(rng.standard_normal((s["n"], 3))) for s in subs]
    for it in range(150):
        if it % 3 == 0:
            A.gmv(g)                                                  # uneven load between the solves
        x = A.local_solve(f)
        for a, b in zip(x, ref):
            assert np.array_equal(a, b), it
    A.destroy()


@pytest.mark.parametrize("cplx", [False, True])
def test_bushes_match_the_level_launches_and_repeat_bitwise(cplx, monkeypatch):
    """the bottom of the tree in one launch per direction (sptrsv16.hip, "bushes": the subtrees whose vectors fit the LDS of a workgroup)
    against the level launches of the same engine (HPDDM_HIP_BUSH16=-1) and against SuperLU; the hand-over inside a bush is ordered
    (supernode after supernode of a round): repeated solves are bitwise equal"""
    n = 14
    K = _laplace3d(n, n, n + 3) * float(n * n)
    N = K.shape[0]
    A = (K - 0.03 * sp.diags(K.diagonal()) + 0.03j * sp.diags(K.diagonal())).tocsr() if cplx else K
    M = sp.tril(A, format="csr")
    M.sort_indices()
    rng = np.random.default_rng(3)
    mu = 8 if cplx else 16
    b = np.asfortranarray(rng.standard_normal((N, mu)) + (1j * rng.standard_normal((N, mu)) if cplx else 0.0))
    xs, nb = [], []
    for bush in ("3", "-1", "13"):
        monkeypatch.setenv("HPDDM_HIP_BUSH16", bush)
        S = hpddm.Subdomain()
        S.numfact(N, M.indptr, M.indices, M.data.astype(np.complex128) if cplx else M.data, sym=True, spd=not cplx)
        x = S.solve(b)
        for _ in range(20):
            assert np.array_equal(S.solve(b), x)
        nb.append(S.info()["bushes"])
        xs.append(x)
        S.destroy()
    assert nb[0] > 0 and nb[1] == 0 and nb[2] > 0 and nb[2] <= nb[0], nb   # taller bushes are fewer
    ref = spl.splu(A.tocsc().astype(np.complex128 if cplx else np.float64)).solve(np.asarray(b))
    for x in xs:
        assert np.abs(x - ref).max() <= 1e-10 * np.abs(ref).max()
    assert np.abs(xs[0] - xs[1]).max() <= 1e-12 * np.abs(ref).max() and np.abs(xs[2] - xs[1]).max() <= 1e-12 * np.abs(ref).max()
