"""complex128 local solver (fp64 + complex128): HpddmHipSubdomainNumfactZ / SolveZ -- native complex panels
(16 bytes per entry, complex LDL^T / LU on the host, the SpTRSV streams the (re, im) pairs with the real tile kernels on the two
planes of every right-hand side) -- against SciPy's complex SuperLU on Helmholtz-like matrices."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spl

from hpddm_amd import hpddm

pytestmark = pytest.mark.gpu


def _laplace3d(n):
    e = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(n, n))
    I = sp.identity(n)
    return (sp.kron(sp.kron(e, I), I) + sp.kron(sp.kron(I, e), I) + sp.kron(sp.kron(I, I), e)).tocsr() * float(n * n)


def _check(A, sym_storage=False, spd=False, mu=1, tol=1e-9):
    n = A.shape[0]
    M = sp.tril(A, format="csr") if sym_storage else A.tocsr()
    M.sort_indices()
    S = hpddm.Subdomain()
    S.numfact(n, M.indptr, M.indices, M.data.astype(np.complex128), sym=sym_storage, spd=spd)
    rng = np.random.default_rng(4)
    b = rng.random((n, mu)) + 1j * rng.random((n, mu))
    b = np.asfortranarray(b if mu > 1 else b[:, 0])
    x = S.solve(b)
    ref = spl.splu(A.tocsc().astype(np.complex128)).solve(np.asarray(b))
    assert np.abs(x - ref).max() <= tol * np.abs(ref).max()
    assert np.linalg.norm(A @ x - b) <= 1e-8 * np.linalg.norm(b)  # examples/solver.py:47
    kind = S.info()["kind"]
    S.destroy()
    return kind


def test_helmholtz_complex_symmetric_full_and_lower_storage():
    n = 12
    K = _laplace3d(n)
    k2 = (2.5 * np.pi) ** 2
    A = (K - k2 * sp.identity(n ** 3) + 1j * 0.8 * k2 * sp.identity(n ** 3)).tocsr()  # shifted Laplacian, complex symmetric, not Hermitian
    assert _check(A) == 1                      # complex symmetric: L D L^T with plain transposes (detected from the values)
    assert _check(A, sym_storage=True, mu=3) == 1
    assert _check(A, sym_storage=True, mu=5) == 1   # 5 complex right-hand sides = 10 real columns inside: blocks of 8 + 2
    assert _check(A, sym_storage=True, mu=8) == 1   # the block of configs[4]: 16 real columns, ONE sweep (wide and narrow tiles)
    assert _check(A, mu=11) == 1                    # 22 real columns: 16 + 4 + 2


def test_strongly_imaginary_diagonal():
    """diagonal with a tiny real part: harmless in complex arithmetic (the pivots are large in modulus)"""
    n = 10
    K = _laplace3d(n)
    A = (1e-6 * K + 1j * (K + 50.0 * sp.identity(n ** 3))).tocsr()
    _check(A, mu=2)


def test_hermitian_and_general_complex_matrices_take_lu():
    n = 9
    K = _laplace3d(n)
    G = sp.random(n ** 3, n ** 3, density=2e-3, random_state=7, format="csr")
    H = (K + 1j * 0.3 * float(n * n) * (G - G.T)).tocsr()   # Hermitian: real symmetric + i * skew
    assert abs(H - H.getH()).max() < 1e-12
    assert _check(H, spd=True, mu=2) == 2      # Hermitian is not complex symmetric: LU (complex Cholesky is not built)
    nonsym = (K + 0.2 * sp.triu(K, 1) + 1j * 0.1 * float(n * n) * G).tocsr()
    assert _check(nonsym, mu=4) == 2           # general complex: LU, 4 right-hand sides = the MFMA forward tiles with 8 real columns


def _helmholtz3d(N, parts, overlap, shift):
    """3-D shifted Laplacian with absorption, split like configs[4] (BASELINE.json): the 7-point stencil of generate3d with the
    diagonal times `shift` (complex, |shift| < 1: indefinite real part), complex right-hand sides, and per subdomain a
    plane-wave coarse space (constant + two complex exponentials), the textbook stand-in for the DtN vectors"""
    from hpddm_amd.generate import generate3d
    subs = generate3d(N, parts, overlap, sym=False, rhs="smooth")
    rng = np.random.default_rng(5)
    out, Z = [], []
    for r, sd in enumerate(subs):
        sd = dict(sd)
        a = sd["a"].astype(np.complex128)
        ia, ja = sd["ia"], sd["ja"]
        for i in range(sd["n"]):
            p = ia[i] + np.nonzero(ja[ia[i]:ia[i + 1]] == i)[0][0]
            a[p] *= shift
        sd["a"] = a
        out.append(sd)
        t = np.arange(sd["n"], dtype=np.float64)
        Z.append(np.stack([np.ones(sd["n"], dtype=np.complex128), np.exp(0.21j * t), np.exp(-0.13j * t + 0.4j * r)], axis=1))
    return out, Z


@pytest.mark.gpu
@pytest.mark.parametrize("correction", ["deflated", "balanced"])
def test_helmholtz_like_two_level_block_gmres_against_oracle(correction):
    """configs[4] of BASELINE.json scaled down: complex 3-D operator on 8 subdomains, two-level RAS with complex deflation
    vectors, Block GMRES with 8 right-hand sides -- device path against the numpy oracle (which is pinned on the reference's
    complex build, tests/test_oracle_golden.py)"""
    from hpddm_amd import hpddm
    from oracle import ras_oracle as ro
    subs, Z = _helmholtz3d(12, 8, 1, 0.8 + 0.02j)   # one restart: 23 / 22 iterations
    mu = 8
    rng = np.random.default_rng(17)
    A, d = hpddm.schwarz_from_subdomains(subs, options=f"-hpddm_schwarz_coarse_correction {correction} -hpddm_krylov_method bgmres -hpddm_gmres_restart 12")
    assert A.complex
    for s, z in enumerate(Z):
        A.set_vectors(s, z)
    A.build_coarse_operator()
    A.call_numfact()
    orc = ro.Oracle(subs, correction=correction)
    orc.multiplicity_scaling([s["d"] for s in subs])
    orc.numfact()
    orc.set_vectors(Z)
    orc.build_coarse(lapacktr=False)   # the plain E, the library's default
    f = orc.exchange([rng.standard_normal((sd["n"], mu)) + 1j * rng.standard_normal((sd["n"], mu)) for sd in subs])

    def close(a, b, tol, what):
        sc = max(np.abs(x).max() for x in b)
        err = max(np.abs(x - y).max() for x, y in zip(a, b)) / sc
        assert err <= tol, (what, err)

    close(A.gmv(f), orc.gmv(f), 1e-13, "GMV")
    close(A.local_solve(f), orc.local_solve(f), 1e-10, "local solve")
    close(A.deflation(f), orc.deflation(f), 1e-9, "deflation")
    close(A.apply(f), orc.apply(f), 1e-9, "apply")
    it, sol, hist = A.solve(f, history=True)
    it_o, sol_o, hist_o = ro.bgmres(orc, f, restart=12)
    assert it == it_o and it < 60
    assert np.allclose(hist, [h[1] for h in hist_o], rtol=1e-4)
    close(sol, sol_o, 1e-7, "solution")
    res = A.compute_residual(sol, f)
    assert np.all(res[1::2] <= 2e-6 * res[0::2] * 50)   # right-preconditioned: true residual within the usual factor of the estimate
    A.destroy()


@pytest.mark.parametrize("where", ["host", "device"])
def test_complex_factorisation_upper_levels_on_the_device(where, monkeypatch):
    """the upper levels of the complex factorisation on the device (numeric_device.hip with T = (re, im) pairs: the real MFMA GEMM on
    the embedded load of B, complex tile kernels with the host's pivot rule) against the host levels only and SuperLU -- complex
    symmetric (L D L^T, plain transposes) and general complex (LU), wide separators, 1 / 3 / 8 right-hand sides"""
    monkeypatch.setenv("HPDDM_HIP_DEVICE_MIN_H", "96" if where == "device" else "100000")
    n = 18
    K = _laplace3d(n)
    N = n ** 3
    k2 = (2.5 * np.pi) ** 2
    A = (K - k2 * sp.identity(N) + 1j * 0.8 * k2 * sp.identity(N)).tocsr()
    assert _check(A, sym_storage=True, mu=1) == 1
    assert _check(A, mu=8) == 1
    G = sp.random(N, N, density=1e-3, random_state=3, format="csr")
    B = (K + 0.2 * sp.triu(K, 1) + 1j * 0.1 * float(n * n) * G).tocsr()
    assert _check(B, mu=3) == 2


def test_complex_device_levels_refuse_a_collapsed_pivot(monkeypatch):
    """the pivot rule of the complex tile kernels: a zero diagonal block of a complex symmetric matrix breaks down loudly on the device
    levels as it does on the host"""
    monkeypatch.setenv("HPDDM_HIP_DEVICE_MIN_H", "64")
    monkeypatch.setenv("HPDDM_HIP_NO_LU_FALLBACK", "1")   # (with it the matrix goes on as LU with pivoting inside the tiles: tests/test_pivoting.py)
    n = 12
    K = _laplace3d(n)
    N = n ** 3
    A = sp.lil_matrix((K * (1.0 + 0.5j)).tocsr())
    A.setdiag(0.0)                                  # zero pivots everywhere: L D L^T without pivoting cannot go on
    S = hpddm.Subdomain()
    M = A.tocsr()
    M.sort_indices()
    with pytest.raises(hpddm.HpddmHipError):
        S.numfact(N, M.indptr, M.indices, M.data.astype(np.complex128), sym=False)
    S.destroy()


@pytest.mark.gpu
def test_cg_on_a_hermitian_positive_definite_operator():
    """IterativeMethod::CG for K = std::complex<double> (include/HPDDM_CG.hpp:31-168): every coefficient of the reference's method
    is the REAL part of a dot product, so the device's CG on the (re, im) arrays is the complex method -- a Hermitian positive
    definite operator (the 7-point stencil with phases e^{+-0.4i} on its off-diagonal entries), ASM, 3 right-hand sides, against
    the oracle's CG and the direct solution"""
    from hpddm_amd import hpddm
    from hpddm_amd.generate import generate3d
    from oracle import ras_oracle as ro
    subs = []
    for sd in generate3d(10, 8, 1, sym=False, rhs="smooth"):
        sd = dict(sd)
        a = sd["a"].astype(np.complex128)
        rows = np.repeat(np.arange(sd["n"]), np.diff(sd["ia"]))
        a *= np.exp(0.4j * np.sign(sd["ja"] - rows))    # local numbering keeps the global order: the blocks of ONE Hermitian matrix
        sd["a"] = a
        subs.append(sd)
    mu = 3
    rng = np.random.default_rng(23)
    A, d = hpddm.schwarz_from_subdomains(subs, options="-hpddm_schwarz_method asm -hpddm_krylov_method cg -hpddm_tol 1e-8 -hpddm_max_it 200")
    assert A.complex
    A.call_numfact()
    orc = ro.Oracle(subs, method="asm")
    orc.multiplicity_scaling([s["d"] for s in subs])
    orc.numfact()
    f = orc.exchange([rng.standard_normal((sd["n"], mu)) + 1j * rng.standard_normal((sd["n"], mu)) for sd in subs])
    it, sol, hist = A.solve(f, history=True)
    it_o, sol_o, hist_o = ro.cg(orc, f, tol=1e-8, max_it=200)
    # the same method step for step: the residual histories agree to 1e-5 as far as both run; the COUNT may differ by one when a
    # residual sits on the threshold to rounding (round 5: the condensed leaves changed the last bits of the local solves, 21 against 20)
    assert abs(it - it_o) <= 1 and 5 < it < 120, (it, it_o)
    ho = np.array([h[1] for h in hist_o])
    m = min(len(hist), len(ho))
    assert m >= min(it, it_o) and np.allclose(np.asarray(hist)[:m], ho[:m], rtol=1e-5)
    sc = max(np.abs(x).max() for x in sol_o)
    assert max(np.abs(x - y).max() for x, y in zip(sol, sol_o)) <= 1e-7 * sc
    r = [ff - g for ff, g in zip(f, orc.gmv(sol))]        # the true residual of the device's solution
    assert max(np.abs(x).max() for x in r) <= 1e-5 * max(np.abs(x).max() for x in f)
    A.destroy()


def test_custom_operator_callbacks_with_complex_scalars():
    """HpddmCustomOperatorSolve for K = std::complex<double> (interface/hpddm_c.cpp:227-230 builds it for every K): GMRES and Block GMRES
    (with and without right-hand-side deflation) on callbacks that see complex host vectors -- a complex symmetric tridiagonal operator
    with its Jacobi preconditioner -- against a direct solve"""
    n, mu = 120, 3
    diag = (np.arange(n) + 2.0) * (1.0 + 0.3j)
    T = sp.diags([-0.5 * np.ones(n - 1), diag, (-0.5 + 0.1j) * np.ones(n - 1)], [-1, 0, 1], format="csr")

    def mv(x, y):
        assert x.dtype == np.complex128
        y[:] = T @ x

    def pc(x, y):
        y[:] = x / diag[:, None]

    eye = sp.identity(n, format="csr", dtype=np.complex128)
    A = hpddm.Schwarz(1)
    A.set_subdomain(0, n, eye.indptr, eye.indices, eye.data, False, [], [])
    A.initialize([np.ones(n)])
    A.set_custom_operator(mv, pc)
    rng = np.random.default_rng(11)
    b = np.asfortranarray(rng.random((n, mu)) + 1j * rng.random((n, mu)))
    bdep = b.copy(order="F")
    bdep[:, 2] = b[:, 0] - 2.0j * b[:, 1]       # a dependent right-hand side for the deflation
    for opts in ("-hpddm_krylov_method gmres", "-hpddm_krylov_method bgmres", "-hpddm_krylov_method bgmres -hpddm_deflation_tol 1e-8",
                 "-hpddm_krylov_method gmres -hpddm_variant left"):
        A.set_option("deflation_tol", -1.0)
        A.option_parse(opts + " -hpddm_tol 1e-9")
        rhs = bdep if "deflation_tol" in opts else b
        exact = spl.spsolve(T.tocsc(), rhs)
        it, sol = A.solve([rhs])
        assert 0 < it <= 20, (opts, it)
        assert np.abs(sol[0] - exact).max() <= 1e-7 * np.abs(exact).max(), opts
    A.destroy()
