"""Pivoting inside the diagonal tiles (the reference's local solvers pivot, include/HPDDM_MUMPS.hpp:228-291): LU
with threshold partial pivoting among the rows of a 64-column tile -- host levels (dense_host.hpp) and device levels
(k_getf2_inv) --, the forward sweep on panels whose top block has dense diagonal tiles (SnDesc::tgs; VALU and MFMA paths), the
fall-back ladder Cholesky -> L D L^T -> LU, zero diagonal entries paired with a neighbour by the ordering.  Against SciPy's
SuperLU at 1e-9; what static pivoting cannot factorise is still refused, loudly."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spl

from hpddm_amd import hpddm
from hpddm_amd._lib import HpddmHipError
from test_library_cpu import _poisson3d, _row_swapped_pairs, _stokes2d

pytestmark = pytest.mark.gpu


def _solve_and_compare(A, mus=(1,), tol=1e-9, sym_storage=False, cplx=False):
    A = A.tocsr()
    n = A.shape[0]
    M = sp.tril(A, format="csr") if sym_storage else A
    M.sort_indices()
    S = hpddm.Subdomain()
    S.numfact(n, M.indptr, M.indices, M.data.astype(np.complex128) if cplx else M.data, sym=sym_storage)
    lu = spl.splu(A.tocsc().astype(np.complex128 if cplx else np.float64))
    rng = np.random.default_rng(7)
    for mu in mus:
        b = rng.random((n, mu)) + (1j * rng.random((n, mu)) if cplx else 0.0)
        b = np.asfortranarray(b if mu > 1 else b[:, 0])
        x = S.solve(b)
        ref = lu.solve(np.asarray(b))
        assert np.abs(x - ref).max() <= tol * np.abs(ref).max(), (mu, np.abs(x - ref).max() / np.abs(ref).max())
    info = S.info()
    swapped = int(np.count_nonzero(S.export("tgs")))
    S.destroy()
    return info["kind"], swapped


def test_matrices_that_used_to_be_refused_are_solved():
    assert _solve_and_compare(sp.csr_matrix(np.array([[0.0, 1.0], [1.0, 0.0]]))) == (2, 1)
    lap = _poisson3d(4)
    for eps in (1e-18, 1e-11):   # collapsed pivot (breakdown of L D L^T) and small pivot (probe): both end as LU with an exchange
        for blk in (np.array([[eps, 1.0], [1.0, eps]]), np.array([[eps, 1.0], [2.0, eps]])):
            kind, swapped = _solve_and_compare(sp.block_diag([lap, sp.csr_matrix(blk)]), mus=(1, 3))
            assert kind == 2 and swapped == 1


@pytest.mark.parametrize("where", ["host", "device"])
def test_rows_exchanged_in_every_tile(where, monkeypatch):
    """two unknowns per node with their rows exchanged: every tile of every front pivots -- host and device levels, real and complex,
    1 to 16 right-hand sides (wave and block tiles of the VALU paths, the 16-column MFMA engine)"""
    monkeypatch.setenv("HPDDM_HIP_DEVICE_MIN_H", "96" if where == "device" else "100000")
    A = _row_swapped_pairs(13)
    kind, swapped = _solve_and_compare(A, mus=(1, 3, 8, 16, 21))
    assert kind == 2 and swapped > 50
    kind, swapped = _solve_and_compare((A * (1.0 + 0.3j)).tocsr(), mus=(1, 8, 11), cplx=True)
    assert kind == 2 and swapped > 50


@pytest.mark.parametrize("where", ["host", "device"])
def test_saddle_point_matrices(where, monkeypatch):
    monkeypatch.setenv("HPDDM_HIP_DEVICE_MIN_H", "96" if where == "device" else "100000")
    kind, _ = _solve_and_compare(_stokes2d(40), mus=(1, 4))
    assert kind in (1, 2)
    kind, swapped = _solve_and_compare(_stokes2d(96), mus=(1, 16))
    assert kind in (1, 2)
    # 3-D: three Laplacians, discrete divergence
    N = 9
    I = sp.identity(N)
    Tm = sp.diags([-1, 2, -1], [-1, 0, 1], shape=(N, N))
    L = (sp.kron(sp.kron(Tm, I), I) + sp.kron(sp.kron(I, Tm), I) + sp.kron(sp.kron(I, I), Tm)).tocsr()
    D = sp.diags([-1, 1], [0, 1], shape=(N, N))
    B = sp.hstack([sp.kron(sp.kron(I, I), D), sp.kron(sp.kron(I, D), I), sp.kron(sp.kron(D, I), I)]).tocsr()[:-1]
    A = sp.bmat([[sp.block_diag([L, L, L]), B.T], [B, None]]).tocsr()
    _solve_and_compare(A, mus=(1, 5))
    _solve_and_compare(A, mus=(2,), sym_storage=True)   # lower triangle only, zero diagonal entries stored or not


@pytest.mark.parametrize("where", ["host", "device"])
def test_indefinite_helmholtz_with_little_absorption(where, monkeypatch):
    """shift (1.0 + 0.005i) k^2: an indefinite complex symmetric operator (the tests of the complex path use 0.8i)"""
    monkeypatch.setenv("HPDDM_HIP_DEVICE_MIN_H", "96" if where == "device" else "100000")
    n = 16
    K = _poisson3d(n) * float(n * n)
    k2 = (3.5 * np.pi) ** 2
    A = (K - (1.0 + 0.005j) * k2 * sp.identity(n ** 3)).tocsr()
    _solve_and_compare(A, mus=(1, 8), cplx=True)
    _solve_and_compare(A, mus=(3,), cplx=True, sym_storage=True)


def test_iterative_refinement_when_the_factor_is_not_backward_stable(monkeypatch):
    """a factor the probe solve of numfact does not accept (growth: here a pivot of 1e-9 in an L D L^T kept on purpose,
    HPDDM_HIP_NO_LU_FALLBACK) but whose error contracts is kept together with the matrix, and every solve through the handle takes the
    refinement steps the probe needed -- what MUMPS / PARDISO do behind Solver::solve after perturbed pivots (include/HPDDM_MUMPS.hpp:
    304-317); HPDDM_HIP_NO_REFINE: refused as before"""
    lap = _poisson3d(4)
    M = sp.block_diag([lap, sp.csr_matrix(np.array([[1e-9, 1.0], [1.0, 1e-9]]))]).tocsr()
    L = sp.tril(M, format="csr")
    L.sort_indices()
    n = M.shape[0]
    monkeypatch.setenv("HPDDM_HIP_NO_LU_FALLBACK", "1")
    monkeypatch.setenv("HPDDM_HIP_NO_REFINE", "1")
    S = hpddm.Subdomain()
    with pytest.raises(HpddmHipError, match="backward stable"):
        S.numfact(n, L.indptr, L.indices, L.data, sym=True)
    S.destroy()
    monkeypatch.delenv("HPDDM_HIP_NO_REFINE")
    lu = spl.splu(M.tocsc())
    rng = np.random.default_rng(5)
    for cplx in (False, True):
        S = hpddm.Subdomain()
        S.numfact(n, L.indptr, L.indices, L.data.astype(np.complex128) * (1.0 + 0.0j) if cplx else L.data, sym=True)
        assert S.info()["kind"] == 1 and 1 <= S.refine_steps() <= 3, (S.info()["kind"], S.refine_steps())
        for mu in (1, 3):
            b = rng.random((n, mu)) + (1j * rng.random((n, mu)) if cplx else 0.0)
            b = np.asfortranarray(b if mu > 1 else b[:, 0])
            x = S.solve(b)
            ref = spl.splu(M.tocsc().astype(np.complex128)).solve(np.asarray(b)) if cplx else lu.solve(np.asarray(b))
            assert np.abs(x - ref).max() <= 1e-9 * np.abs(ref).max(), (cplx, mu, np.abs(x - ref).max() / np.abs(ref).max())
        S.destroy()
    # the rule: a stable factor takes no refinement step
    S = hpddm.Subdomain()
    K = sp.tril(lap, format="csr")
    S.numfact(lap.shape[0], K.indptr, K.indices, K.data, sym=True)
    assert S.refine_steps() == 0
    S.destroy()


@pytest.mark.parametrize("cplx", [False, True])
def test_refinement_inside_the_batched_sweeps_of_an_operator(cplx, monkeypatch):
    """the local solves of a Schwarz operator refine as well (Schwarz::solve_factor: the batched sweep, then the steps of the subdomains
    that need them through their own plans, then the partition of unity).  Forced here on factors that do not need it
    (HPDDM_HIP_FORCE_REFINE, a developer aid): the apply and the Krylov solve of the operator are those of the plain sweeps."""
    from hpddm_amd.generate import generate3d, generate_helmholtz3d
    if cplx:
        subs = generate_helmholtz3d(12, 8, wavenumber=2.0 * np.pi * 3.0)
        build = lambda: hpddm.schwarz_from_subdomains(subs, options="-hpddm_schwarz_method oras", multiplicity=False)
    else:
        subs = generate3d(12, 8, 1, sym=True, rhs="smooth")
        build = lambda: hpddm.schwarz_from_subdomains(subs)   # (no -hpddm_operator_spd: L D L^T, the kind the probe looks at)
    outs = []
    for force in (None, "2"):
        if force:
            monkeypatch.setenv("HPDDM_HIP_FORCE_REFINE", force)
        A, d = build()
        if cplx:
            for s_, sd in enumerate(subs):
                A.set_optimized_matrix(s_, sd["n"], sd["ia"], sd["ja"], sd["a_opt"], False)
        A.call_numfact()
        if cplx:
            rng = np.random.default_rng(4)
            f = [np.asfortranarray(rng.standard_normal((sd["n"], 2)) + 1j * rng.standard_normal((sd["n"], 2))) for sd in subs]
        else:
            f = [s["f"] for s in subs]
        ap = A.apply(f)
        it, sol = A.solve(f)
        outs.append((ap, it, sol, A.subdomain(0).refine_steps()))
        A.destroy()
    (a0, it0, s0, r0), (a1, it1, s1, r1) = outs
    assert r0 == 0 and r1 == 2 and it0 == it1
    scale = max(np.abs(v).max() for v in a0)
    assert max(np.abs(x - y).max() for x, y in zip(a0, a1)) <= 1e-12 * scale
    assert max(np.abs(x - y).max() for x, y in zip(s0, s1)) <= 1e-5 * max(np.abs(v).max() for v in s0)   # (two Krylov solves to 1e-6 on applies that differ in the last bits)


def test_perturbed_pivots_for_a_tile_without_a_usable_one(monkeypatch):
    """a supernode whose diagonal tile holds no usable pivot -- MUMPS / PARDISO delay such a pivot to an ancestor, the static structure
    cannot --: the last rung of numfact factorises once more as LU on the host with the pivot REPLACED by +-sqrt(eps) max |a_ij|
    (static pivoting), and the probe solve with its iterative refinement decides whether that serves.  Supernodes of ONE column
    (leaf_size = 1) so that a vertex with a collapsed diagonal entry has no row to exchange with; against SuperLU at 1e-9.
    HPDDM_HIP_NO_PERTURB: refused as before.  A singular matrix is still refused (next test)."""
    A = _poisson3d(4).tocsr()
    A.sort_indices()
    n = A.shape[0]
    S = hpddm.Subdomain(leaf_size=1)   # the supernodes of this pattern (the ordering sees the pattern only): one of a single column
    S.numfact(n, A.indptr, A.indices, A.data, sym=False)
    blk, perm = S.export("blk_ptr"), S.export("perm")
    S.destroy()
    single = [k for k in range(len(blk) - 1) if blk[k + 1] - blk[k] == 1]
    assert single, "no supernode of one column"
    v = int(perm[blk[single[0]]])
    A = A.tolil()
    A[v, v] = 1e-20
    A = A.tocsr()
    A.sort_indices()
    lu = spl.splu(A.tocsc())
    monkeypatch.setenv("HPDDM_HIP_NO_PERTURB", "1")
    S = hpddm.Subdomain(leaf_size=1)
    with pytest.raises(HpddmHipError, match="pivot"):
        S.numfact(n, A.indptr, A.indices, A.data, sym=False)
    S.destroy()
    monkeypatch.delenv("HPDDM_HIP_NO_PERTURB")
    S = hpddm.Subdomain(leaf_size=1)
    S.numfact(n, A.indptr, A.indices, A.data, sym=False)
    assert S.info()["kind"] == 2 and 1 <= S.refine_steps() <= 3, (S.info()["kind"], S.refine_steps())
    rng = np.random.default_rng(9)
    for mu in (1, 4):
        b = np.asfortranarray(rng.random((n, mu)) if mu > 1 else rng.random(n))
        x = S.solve(b)
        ref = lu.solve(np.asarray(b))
        assert np.abs(x - ref).max() <= 1e-9 * np.abs(ref).max(), (mu, np.abs(x - ref).max() / np.abs(ref).max())
    S.destroy()


def test_what_static_pivoting_cannot_do_is_refused():
    lap = _poisson3d(4)
    for blk in (np.array([[1.0, 1.0], [1.0, 1.0]]), np.array([[0.0, 0.0], [1.0, 2.0]])):   # singular tiles
        M = sp.block_diag([lap, sp.csr_matrix(blk)]).tocsr()
        M.sort_indices()
        S = hpddm.Subdomain()
        with pytest.raises(HpddmHipError, match="pivot"):
            S.numfact(M.shape[0], M.indptr, M.indices, M.data, sym=False)
        S.destroy()
    # large entries outside the tiles (other supernodes): the factor grows, the probe solve that closes numfact says so
    A = (_poisson3d(8) + sp.random(512, 512, density=0.01, random_state=3) * 5.0).tolil()
    A.setdiag(A.diagonal() * 0.01)
    A = A.tocsr()
    S = hpddm.Subdomain()
    try:
        S.numfact(512, A.indptr, A.indices, A.data, sym=False)
    except HpddmHipError as e:
        assert "pivot" in str(e)
    else:   # never a silently wrong solution: whatever is accepted is accurate
        b = np.ones(512)
        assert np.abs(A @ S.solve(b) - b).max() < 1e-7 * np.abs(b).max() * abs(A).sum(axis=1).max()
    S.destroy()


def test_inertia_from_the_ldlt_factor():
    """Solver::inertia (include/HPDDM_MUMPS.hpp:292-302, MUMPS' INFOG(12)): the number of negative pivots of L D L^T = the number of
    negative eigenvalues (Sylvester), against numpy.linalg.eigvalsh; 0 for a Cholesky factor; -3 when the matrix went through LU"""
    import scipy.sparse as sp
    from hpddm_amd import hpddm
    n1 = 9
    e = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(n1, n1))
    I = sp.identity(n1)
    rng = np.random.default_rng(3)
    K = (sp.kron(sp.kron(e, I), I) + sp.kron(sp.kron(I, e), I) + sp.kron(sp.kron(I, I), e) + sp.diags(0.3 * rng.random(n1 ** 3))).tocsr()   # (no symmetry left: no leaf block is singular at a shift)
    lam = np.linalg.eigvalsh(K.toarray())
    seen = set()
    for k in (None, 3, 40, 300):
        shift = 0.0 if k is None else lam[k] + 0.37 * (lam[k + 1] - lam[k])
        A = (K - shift * sp.identity(n1 ** 3)).tocsr()
        M = sp.tril(A, format="csr")
        M.sort_indices()
        S = hpddm.Subdomain()
        S.numfact(A.shape[0], M.indptr, M.indices, M.data, sym=True, spd=False)
        expect = 0 if k is None else k + 1
        got = S.inertia()
        kind = S.info()["kind"]
        seen.add(kind)
        assert (kind == 1 and got == expect) or (kind == 2 and got == -3), (k, kind, got, expect)
        S.destroy()
    assert 1 in seen, "no shift went through L D L^T"
    S = hpddm.Subdomain()
    M = sp.tril(K, format="csr")
    S.numfact(K.shape[0], M.indptr, M.indices, M.data, sym=True, spd=True)
    assert S.info()["kind"] == 0 and S.inertia() == 0          # Cholesky
    S.destroy()


def test_geneo_estimate_nu_counts_the_eigenvalues_below_the_threshold():
    """-hpddm_geneo_estimate_nu (include/HPDDM_schwarz.hpp:686-703): nu = inertia(A_N - threshold B) instead of a guess"""
    from hpddm_amd import hpddm
    from hpddm_amd.generate import generate3d
    from oracle.ras_oracle import Oracle, csr_full
    subs = generate3d(12, 8, 2, sym=True, rhs="smooth", neumann=True)
    thr = 0.85   # eigenvalues of these pencils: 0.5113, 0.7685 (double), 0.8111 | 0.8708 ...: four below the threshold
    A, d = hpddm.schwarz_from_subdomains(subs, options=f"-hpddm_operator_spd -hpddm_geneo_nu 3 -hpddm_geneo_threshold {thr} -hpddm_geneo_estimate_nu 1 -hpddm_eigensolver_tol 1e-9")
    orc = Oracle(subs)
    orc.multiplicity_scaling([s["d"] for s in subs])
    neumann = [csr_full(dict(sd, a=sd["a_neumann"])) for sd in subs]
    ref = orc.geneo(neumann, 20)
    for s, sd in enumerate(subs):
        below = int((ref[s] <= thr).sum())
        assert below == 4
        lam = A.solve_gevp(s, sd["n"], sd["ia"], sd["ja"], sd["a_neumann"], sd["sym"])
        assert len(lam) == max(1, below), (s, len(lam), below, ref[s])
        assert np.all(np.abs(lam - ref[s][:len(lam)]) <= 1e-6 * np.maximum(np.abs(ref[s][:len(lam)]), 1e-3))
    A.destroy()
