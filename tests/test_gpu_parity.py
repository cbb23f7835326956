"""Parity of the HIP path (through the C ABI) with the golden vectors of the compiled reference and with the oracle.

Tolerances (fp64): vectors within 1e-10 relative per apply; iteration counts equal; residual
histories within the 7 printed digits of the reference log.
"""
import numpy as np
import pytest

import golden_util as gu
from hpddm_amd import hpddm
from hpddm_amd.generate import generate2d, generate3d
from oracle.ras_oracle import Oracle

pytestmark = pytest.mark.gpu


def _close(a, b, rtol, what):
    scale = max(np.abs(np.concatenate([np.ravel(x) for x in b])).max(), 1e-300)
    err = max(np.abs(np.ravel(x) - np.ravel(y)).max() for x, y in zip(a, b)) / scale
    assert err <= rtol, f"{what}: relative error {err:.3e} > {rtol:.1e}"


def _build(g, subs):
    hpddm.require_device()
    A, d = hpddm.schwarz_from_subdomains(subs, options=gu.hpddm_args(g))
    opt = gu.options(g)
    if opt["correction"]:
        for s, Z in enumerate(gu.deflation_vectors(g, subs)):
            A.set_vectors(s, Z)
        if A.complex and any(len(sd["neighbors"]) != len(subs) - 1 for sd in subs):
            # the fixtures come from the reference built with its dense LapackTR coarse back-end, which factorises E^T when
            # the coarse matrix is not full (include/HPDDM_LAPACK.hpp:417, :348-352): reproduce that build
            A.set_option("hip_coarse_transpose", 1)
        A.build_coarse_operator()
    if "a_opt_r0" in g:   # callNumfact(A_opt): ORAS / SORAS with an optimised local matrix
        for s, t in enumerate(gu.optimized_matrices(g, subs)):
            A.set_optimized_matrix(s, t["n"], t["ia"], t["ja"], t["a"], t["sym"])
    A.call_numfact()
    return A, d, opt


@pytest.mark.parametrize("name", gu.SMALL_CASES + gu.OPTIMIZED_CASES + gu.PENALIZED_CASES + gu.MULTI_VECTOR_CASES + gu.COMPLEX_CASES
                         + gu.COMPLEX_BGMRES_CASES)
def test_functions_match_reference(name):
    g = gu.load(name)
    subs = gu.subdomains(g)
    A, d, opt = _build(g, subs)
    for r in range(len(subs)):
        assert np.abs(d[r] - g[f"d_r{r}"]).max() <= 1e-15, "multiplicityScaling"
    f = gu.vecs(g, "f")
    _close(A.exchange(f), gu.vecs(g, "exchange_out"), 1e-14, "exchange")
    _close(A.gmv(f), gu.vecs(g, "gmv_out"), 1e-13, "GMV")
    _close(A.local_solve(f), gu.vecs(g, "solve_out"), 1e-10, "Solver::solve")
    loose = 1e-7 if "nu3" in name else 1e-10   # three smooth vectors per subdomain: the coarse matrix is ill-conditioned
    _close(A.apply(f), gu.vecs(g, "apply_out"), loose, "apply")
    if opt["correction"]:
        _close(A.deflation(f), gu.vecs(g, "deflation_out"), loose, "deflation")
    A.destroy()


@pytest.mark.parametrize("name", gu.SMALL_CASES + gu.OPTIMIZED_CASES + gu.MULTI_VECTOR_CASES + gu.COMPLEX_CASES)
def test_gmres_matches_reference(name):
    g = gu.load(name)
    subs = gu.subdomains(g)
    A, d, opt = _build(g, subs)
    f = gu.vecs(g, "f")
    it, sol, hist = A.solve(f, history=True)
    assert it == int(g["iterations_r0"][0])
    ref = g["history"]
    assert len(hist) == len(ref)
    assert np.all(np.abs(hist - ref[:, 1]) <= (5e-3 if "nu3" in name else 2e-6) * ref[:, 1])
    _close(sol, gu.vecs(g, "sol"), 1e-5 if "nu3" in name else 1e-8, "solution")
    assert np.allclose(A.compute_residual(sol, f), g["residual_r0"], rtol=1e-3 if "nu3" in name else 1e-5)
    if "residual_l1_r0" in g:   # the other norms of Schwarz::computeResidual, on the reference's own solution
        rs, rf = gu.vecs(g, "sol"), gu.vecs(g, "f")
        assert np.allclose(A.compute_residual(rs, rf, "l1"), g["residual_l1_r0"], rtol=1e-6)
        assert np.allclose(A.compute_residual(rs, rf, "linfty"), g["residual_linfty_r0"], rtol=1e-6)
    A.destroy()


def test_flexible_gmres_matches_reference():
    """-hpddm_variant flexible (include/HPDDM_GMRES.hpp:116-117,139): restart 8, two right-hand sides, 27 iterations"""
    g = gu.load("p40_fgmres_restart8_mu2")
    subs = gu.subdomains(g)
    A, d, opt = _build(g, subs)
    f = gu.vecs(g, "f")
    it, sol, hist = A.solve(f, history=True)
    assert it == int(g["iterations_r0"][0]) == 27
    ref = g["history"]
    assert len(hist) == len(ref) and np.all(np.abs(hist - ref[:, 1]) <= 1e-4 * ref[:, 1])
    _close(sol, gu.vecs(g, "sol"), 1e-7, "solution")
    assert np.allclose(A.compute_residual(sol, f), g["residual_r0"], rtol=1e-4)
    A.destroy()


def test_config1_45_iterations():
    """BASELINE.json configs[0]: examples/schwarz.cpp 2-D Poisson 200x200, 4 subdomains, one-level RAS -> 45 iterations"""
    g = gu.load("c1_p200_onelevel")
    subs = generate2d(200, 200, 4)
    A, d = hpddm.schwarz_from_subdomains(subs)
    A.call_numfact()
    f = [s["f"] for s in subs]
    _close(A.apply(f), [g[f"apply_out_r{r}"] for r in range(4)], 1e-10, "apply")
    it, sol, hist = A.solve(f, history=True)
    assert it == 45
    assert np.all(np.abs(hist - g["history"][:, 1]) <= 1e-4 * g["history"][:, 1])
    _close(sol, [g[f"sol_r{r}"] for r in range(4)], 1e-8, "solution")
    res = A.compute_residual(sol, f)
    assert np.allclose(res, g["residual_r0"], rtol=1e-5)
    assert res[1] / res[0] <= 1e-2  # the reference's own acceptance test, examples/schwarz.cpp:140-144
    A.destroy()


@pytest.mark.parametrize("N,parts,overlap,sym,mu", [(12, 8, 1, True, 1), (14, 8, 2, True, 3), (10, 4, 1, False, 2), (16, 2, 1, True, 5), (12, 8, 1, True, 4), (10, 4, 1, False, 8)])
def test_3d_against_oracle(N, parts, overlap, sym, mu):
    """3-D Poisson (configs 2-3 family) at sizes the oracle finishes in seconds: every hot-path function + GMRES"""
    subs = generate3d(N, parts, overlap, sym=sym, rhs="smooth")
    A, d = hpddm.schwarz_from_subdomains(subs, options="-hpddm_operator_spd" if sym else "")
    A.call_numfact()
    orc = Oracle(subs)
    dd = orc.multiplicity_scaling([s["d"] for s in subs])
    for a, b in zip(d, dd):
        assert np.abs(a - b).max() <= 1e-15
    orc.numfact()
    rng = np.random.default_rng(7)
    f = [rng.random((s["n"], mu)) if mu > 1 else rng.random(s["n"]) for s in subs]
    f = orc.exchange(f)  # consistent on the overlap
    _close(A.exchange(f), orc.exchange(f), 1e-14, "exchange")
    _close(A.gmv(f), orc.gmv(f), 1e-13, "GMV")
    _close(A.local_solve(f), orc.local_solve(f), 1e-10, "Solver::solve")
    _close(A.apply(f), orc.apply(f), 1e-10, "apply")
    it, sol = A.solve(f)
    it_o, sol_o, _ = orc.gmres(f)
    assert it == it_o
    _close(sol, sol_o, 1e-8, "solution")
    A.destroy()


def test_3d_two_level_against_oracle():
    """deflated two-level apply with several deflation vectors per subdomain (low-order polynomials x partition of unity)"""
    N, parts = 12, 8
    subs = generate3d(N, parts, 2, sym=True, rhs="smooth")
    A, d = hpddm.schwarz_from_subdomains(subs, options="-hpddm_operator_spd -hpddm_schwarz_coarse_correction deflated")
    orc = Oracle(subs, correction="deflated")
    orc.multiplicity_scaling([s["d"] for s in subs])
    Z = []
    for s, sd in enumerate(subs):
        i0, i1, j0, j1, k0, k1 = sd["box"]
        z, y, x = np.meshgrid(np.arange(k0, k1) / N, np.arange(j0, j1) / N, np.arange(i0, i1) / N, indexing="ij")
        Zs = np.stack([np.ones(sd["n"]), x.ravel(), y.ravel(), z.ravel()], axis=1)
        Z.append(np.asfortranarray(Zs))
        A.set_vectors(s, Zs)
    A.build_coarse_operator()
    A.call_numfact()
    orc.set_vectors(Z)
    orc.build_coarse()
    orc.numfact()
    f = orc.exchange([np.random.default_rng(3).random(sd["n"]) for sd in subs])
    _close(A.deflation(f), orc.deflation(f), 1e-10, "deflation")
    _close(A.apply(f), orc.apply(f), 1e-10, "apply")
    it, sol = A.solve(f)
    it_o, sol_o, _ = orc.gmres(f)
    assert it == it_o
    _close(sol, sol_o, 1e-8, "solution")
    A.destroy()


def test_local_solver_kinds_and_properties():
    """Solver concept: Cholesky / LDL^T / LU, 'C' and 'F' numbering, in-place solve, multiple right-hand sides, linearity"""
    import scipy.sparse as sp
    N = 9
    I = sp.identity(N)
    T = sp.diags([-1, 2, -1], [-1, 0, 1], shape=(N, N))
    A0 = (sp.kron(sp.kron(T, I), I) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(I, I), T)).tocsr()
    n = A0.shape[0]
    rng = np.random.default_rng(0)
    cases = {
        "chol": (A0, True, True),
        "ldlt": ((A0 - 1.7 * sp.identity(n)).tocsr(), True, False),
        "lu": ((A0 + sp.diags(rng.random(n)) + 0.3 * sp.triu(A0, 1)).tocsr(), False, False),
    }
    for kind, (M, sym, spd) in cases.items():
        Min = sp.tril(M).tocsr() if sym else M
        Min.sort_indices()
        for numbering in ("C", "F"):
            base = 1 if numbering == "F" else 0
            S = hpddm.Subdomain()
            S.numfact(n, Min.indptr + base, Min.indices + base, Min.data, sym=sym, numbering=numbering, spd=spd)
            for mu in (1, 2, 3, 7, 8, 9):
                b = np.asfortranarray(rng.random((n, mu)))
                x = S.solve(b)
                r = np.abs(M @ x - b).max() / np.abs(b).max()
                assert r < 1e-10, (kind, numbering, mu, r)
            # in place + linearity
            b1, b2 = rng.random(n), rng.random(n)
            x1, x2 = S.solve(b1), S.solve(b2)
            y = 2.0 * b1 - 3.0 * b2
            S.solve(y, y)
            assert np.abs(y - (2 * x1 - 3 * x2)).max() <= 1e-11 * np.abs(y).max()
            S.destroy()


def test_larger_size_properties():
    """size-independent checks at a larger size (no oracle): residual of the direct solve, D-weighted partition of unity,
    exchange idempotence on consistent vectors, GMRES residual threshold of the reference"""
    subs = generate3d(32, 8, 1, sym=True, rhs="smooth")
    A, d = hpddm.schwarz_from_subdomains(subs, options="-hpddm_operator_spd")
    A.call_numfact()
    import scipy.sparse as sp
    f = [s["f"] for s in subs]
    x = A.local_solve(f)
    for sd, xs, fs in zip(subs, x, f):
        M = sp.csr_matrix((sd["a"], sd["ja"], sd["ia"]), shape=(sd["n"], sd["n"]))
        M = M + sp.tril(M, -1).T
        assert np.linalg.norm(M @ xs - fs) / np.linalg.norm(fs) < 1e-11
    ones = [np.ones(s["n"]) for s in subs]
    _close(A.exchange(ones), ones, 1e-14, "partition of unity")   # sum_j R_j^T D_j R_j = I
    it, sol = A.solve(f)
    res = A.compute_residual(sol, f)
    assert it <= 45 and res[1] / res[0] <= 1e-2
    A.destroy()


def test_device_levels_of_the_factorisation():
    """fronts with >= 768 rows are factorised on the device (numeric_device.hip, MFMA f64): a 40^3 subdomain has a
    1600-wide top separator.  Direct-solve residual (examples/schwarz.cpp:178 asks 1e-6) and agreement with SuperLU."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    N = 40
    I = sp.identity(N)
    T = sp.diags([-1, 2, -1], [-1, 0, 1], shape=(N, N))
    A = (sp.kron(sp.kron(T, I), I) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(I, I), T)).tocsr()
    n = A.shape[0]
    L = sp.tril(A).tocsr()
    L.sort_indices()
    S = hpddm.Subdomain()
    S.numfact(n, L.indptr, L.indices, L.data, sym=True, spd=True)
    lu = spl.splu(sp.csc_matrix(A))
    for mu in (2, 8):   # 8 right-hand sides take the MFMA tiles on the wide panels
        b = np.asfortranarray(np.random.default_rng(11).random((n, mu)))
        x = S.solve(b)
        assert np.abs(A @ x - b).max() / np.abs(b).max() < 1e-10
        ref = lu.solve(b)
        assert np.abs(x - ref).max() <= 1e-10 * np.abs(ref).max()
    S.destroy()


@pytest.mark.parametrize("kind", ["chol", "ldlt", "lu"])
def test_device_levels_of_two_factorisations_in_flight(kind, monkeypatch):
    """two scratch slots (HPDDM_HIP_DEVICE_SLOTS=2: the device levels of two factorisations interleave, each on its own streams, upload
    ring and arena) and eight streams per factorisation instead of four: three subdomains factorised from three host threads at once,
    all three kinds -- the same solutions as SuperLU.  (One slot is the default: two measured no faster)"""
    import threading

    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    monkeypatch.setenv("HPDDM_HIP_DEVICE_SLOTS", "2")
    monkeypatch.setenv("HPDDM_HIP_FACTOR_STREAMS", "8")
    N = 32
    I = sp.identity(N)
    T = sp.diags([-1, 2, -1], [-1, 0, 1], shape=(N, N))
    A0 = (sp.kron(sp.kron(T, I), I) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(I, I), T)).tocsr()
    n = A0.shape[0]
    rng = np.random.default_rng(5)
    mats = []
    for s in range(3):
        if kind == "chol":
            M = A0 + sp.diags(0.1 * s + rng.random(n))
        elif kind == "ldlt":
            M = A0 - (0.05 + 0.01 * s) * sp.identity(n)
        else:
            M = A0 + 0.2 * sp.triu(A0, 1) + sp.diags((1 + s) * rng.random(n))
        mats.append(M.tocsr())
    subs, errs = [hpddm.Subdomain() for _ in mats], []

    def work(s):
        try:
            M = mats[s]
            if kind == "lu":
                F = M.copy()
                F.sort_indices()
                subs[s].numfact(n, F.indptr, F.indices, F.data, sym=False, spd=False)
            else:
                L = sp.tril(M).tocsr()
                L.sort_indices()
                subs[s].numfact(n, L.indptr, L.indices, L.data, sym=True, spd=kind == "chol")
        except Exception as e:   # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=work, args=(s,)) for s in range(3)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for s, M in enumerate(mats):
        assert subs[s].info()["kind"] in {"chol": (0,), "ldlt": (1, 2), "lu": (2,)}[kind]     # (an L D L^T that grows falls back to LU)
        b = np.asfortranarray(rng.random((n, 2)))
        x = subs[s].solve(b)
        ref = spl.splu(sp.csc_matrix(M)).solve(b)
        assert np.abs(x - ref).max() <= (1e-6 if kind == "lu" else 1e-9) * np.abs(ref).max()
        subs[s].destroy()


def test_geneo_coarse_space_against_arpack():
    """GenEO: eigenvalues of (A_N, scaleIntoOverlap(A_N)) against scipy's ARPACK (what the reference calls),
    then the deflated two-level operator built on them (invariant under a change of basis of each local space)."""
    N, parts, ov, nu = 14, 8, 2, 6
    subs = generate3d(N, parts, ov, sym=True, rhs="smooth", neumann=True)
    A, d = hpddm.schwarz_from_subdomains(subs, options=f"-hpddm_operator_spd -hpddm_schwarz_coarse_correction deflated -hpddm_geneo_nu {nu} -hpddm_eigensolver_tol 1e-9")
    orc = Oracle(subs, correction="deflated")
    orc.multiplicity_scaling([s["d"] for s in subs])
    neumann = []
    for sd in subs:
        sn = dict(sd)
        sn["a"] = sd["a_neumann"]
        from oracle.ras_oracle import csr_full
        neumann.append(csr_full(sn))
    lam_ref = orc.geneo(neumann, nu)
    for s, sd in enumerate(subs):
        lam = A.solve_gevp(s, sd["n"], sd["ia"], sd["ja"], sd["a_neumann"], sd["sym"])
        assert len(lam) == nu
        assert np.all(np.abs(lam - lam_ref[s]) <= 1e-6 * np.maximum(np.abs(lam_ref[s]), 1e-3)), (s, lam, lam_ref[s])
    A.build_coarse_operator()
    A.call_numfact()
    orc.build_coarse()
    orc.numfact()
    f = orc.exchange([np.random.default_rng(9).random(sd["n"]) for sd in subs])
    _close(A.deflation(f), orc.deflation(f), 1e-6, "deflation on the GenEO space")
    _close(A.apply(f), orc.apply(f), 1e-6, "two-level apply")
    it, sol = A.solve(f)
    it_o, sol_o, _ = orc.gmres(f)
    assert abs(it - it_o) <= 1 and it < 20
    A.destroy()


@pytest.mark.parametrize("method", ["bgmres", "bcg"])
def test_block_methods_take_more_than_eight_right_hand_sides(method):
    """The reference's block methods take any number of right-hand sides (examples run with arbitrary -generate_random_rhs); here
    blocks hold at most 8, so 11 right-hand sides are solved as 8 + 3: every column must meet the tolerance."""
    subs = generate3d(12, 8, 1, sym=True, rhs="smooth")
    A, d = hpddm.schwarz_from_subdomains(subs, options="-hpddm_operator_spd -hpddm_krylov_method " + method + (" -hpddm_schwarz_method asm" if method == "bcg" else ""))
    A.call_numfact()
    rng = np.random.default_rng(7)
    f = A.exchange([rng.random((s["n"], 11)) for s in subs])  # consistent right-hand sides
    it, sol = A.solve(f)
    assert 0 < it < 60
    res = A.compute_residual(sol, f).reshape(11, 2)
    assert np.all(res[:, 1] <= 1e-5 * res[:, 0]), res
    A.destroy()


@pytest.mark.parametrize("name", ["p40_bgmres_mu4", "p40_bgmres_deflated_mu2", "p30_6ranks_bgmres_left_mu3", "p40_fbgmres_mu3",
                                  "p40_bgmres_rhs_deflation_mu4", "p40_bgmres_rhs_deflation_restart_mu4", "p40_bgmres_mgs_qrmgs_mu3",
                                  "p40_bgmres_qrcgs_mu3"] + gu.COMPLEX_BGMRES_CASES)
def test_bgmres_matches_reference(name):
    """Block GMRES: iteration count, residual history and solution of the compiled reference.  The two
    rhs_deflation fixtures run with -hpddm_deflation_tol and a last right-hand side f_0 + 2 f_1: one column is deflated at
    every restart (include/HPDDM_GMRES.hpp:201-205)."""
    g = gu.load(name)
    subs = gu.subdomains(g)
    A, d, opt = _build(g, subs)
    f = gu.vecs(g, "f")
    it, sol, hist = A.solve(f, history=True)
    assert it == int(g["iterations_r0"][0])
    ref = g["history"]
    assert len(hist) == len(ref)
    assert np.all(np.abs(hist - ref[:, 1]) <= 2e-4 * ref[:, 1])  # five restarts amplify round-off in the last entries
    _close(sol, gu.vecs(g, "sol"), 1e-7, "solution")
    assert np.allclose(A.compute_residual(sol, f), g["residual_r0"], rtol=1e-4)
    A.destroy()


@pytest.mark.parametrize("name,its", [("p40_cg_asm", 40), ("z_p30_cg_asm_hpd_mu3", 17)])
def test_cg_matches_reference(name, its):
    """PCG with the (symmetric) additive Schwarz preconditioner: the reference's 40 iterations; and the reference built
    for K = std::complex<double> on a Hermitian positive definite operator with three complex right-hand sides: 17"""
    g = gu.load(name)
    subs = gu.subdomains(g)
    A, d, opt = _build(g, subs)
    f = gu.vecs(g, "f")
    it, sol, hist = A.solve(f, history=True)
    assert it == int(g["iterations_r0"][0]) == its
    ref = g["history"]
    assert len(hist) == len(ref) and np.all(np.abs(hist - ref[:, 1]) <= 1e-4 * ref[:, 1])
    _close(sol, gu.vecs(g, "sol"), 1e-7, "solution")
    _close(A.apply(f), gu.vecs(g, "apply_out"), 1e-10, "ASM apply")
    A.destroy()


@pytest.mark.parametrize("name", ["p40_gcrodr_two_solves", "p40_gcrodr_same_system", "p30_6ranks_gcrodr_left_deflated_mu2", "p40_gcrodr_target_lm",
                                  "p40_gcrodr_cycle_end", "p40_bgcrodr_two_solves_mu2", "p30_6ranks_bgcrodr_left_deflated_mu3",
                                  "z_p30_gcrodr_two_solves", "z_p30_gcrodr_mu2", "z_p30_gcrodr_left_mgs", "z_p30_gcrodr_target_lm_same_system",
                                  "z_p30_bgcrodr_two_solves_mu2", "z_p30_6ranks_bgcrodr_left_mu3"])
def test_gcrodr_matches_reference(name):
    """GCRO-DR (include/HPDDM_GCRODR.hpp:34-443), two successive solves on one operator: the first one builds the recycled
    subspace (harmonic Ritz vectors after its first cycle, generalised eigenproblem at every later restart), the second one
    starts from it -- the reference's 19 then 15 iterations where GMRES(10) needs 24, and its residual histories.  With
    -hpddm_recycle_same_system the subspace is frozen during the second solve, like in the reference.  The bgcrodr fixtures run
    the block method (include/HPDDM_GCRODR.hpp:445-905): 18 then 13 iterations for two right-hand sides; cycle_end is a run whose
    first solve converges on the last step of a cycle (the reference then recycles an un-normalised last vector).  The z_ fixtures
    are the reference built for K = std::complex<double> (33 + 33, 37 + 35 with two right-hand sides, 17 + 15 left-preconditioned
    with MGS, 22 + 22 with target LM and a frozen subspace; block method: 36 + 32 for two right-hand sides, 19 + 18 for three,
    left-preconditioned on 6 subdomains): GCRO-DR and Block GCRO-DR in complex arithmetic, krylov_complex.hip."""
    g = gu.load(name)
    subs = gu.subdomains(g)
    A, d, opt = _build(g, subs)
    f, f2 = gu.vecs(g, "f"), gu.vecs(g, "f2")
    it, sol, hist = A.solve(f, history=True)
    it2, sol2, hist2 = A.solve(f2, history=True)
    assert it == int(g["iterations_r0"][0]) and it2 == int(g["iterations2_r0"][0])
    ref = g["history"][:, 1]
    assert len(ref) == it + it2
    # (non-block method, several right-hand sides: the reference runs them in lock-step and keeps printing the residual of the first one after
    # it has converged, while the others go on; solved one at a time, that tail of the printed history -- not the iterates -- is not reproduced)
    cut = 3 if name == "z_p30_gcrodr_mu2" else 0
    assert np.allclose(hist[:it - cut], ref[:it - cut], rtol=1e-4) and np.allclose(hist2[:it2 - cut], ref[it:it + it2 - cut], rtol=1e-4)
    _close(sol, gu.vecs(g, "sol"), 1e-8, "solution")
    _close(sol2, gu.vecs(g, "sol2"), 1e-8, "second solution")
    A.destroy()


@pytest.mark.parametrize("name", ["p40_bgcrodr_rhs_deflation_mu4", "z_p30_bgcrodr_rhs_deflation_mu4"])
def test_block_gcrodr_with_rhs_deflation_matches_reference(name):
    """Block GCRO-DR with -hpddm_deflation_tol (include/HPDDM_GCRODR.hpp:545-600; round 6: the last Krylov variant the library refused):
    the fourth right-hand side is f_0 + 2 f_1, the RRQR of every residual block finds three columns, the cycles -- block Hessenberg
    matrix, recycled space, harmonic-Ritz and generalised eigenproblems -- run on blocks of three, the deflated right-hand side receives
    the corrections times R11^{-1} R12.  The reference's 17 iterations (real scalars) and 32 (complex), its histories and solutions."""
    g = gu.load(name)
    subs = gu.subdomains(g)
    A, d, opt = _build(g, subs)
    f = gu.vecs(g, "f")
    it, sol, hist = A.solve(f, history=True)
    ref = g["history"][:, 1]
    assert it == int(g["iterations_r0"][0]) == len(ref), (it, len(ref))
    assert np.allclose(hist[:it], ref, rtol=1e-4)
    _close(sol, gu.vecs(g, "sol"), 1e-8, "solution")
    assert np.allclose(A.compute_residual(sol, f), g["residual_r0"], rtol=1e-5)
    # a second solve on the same operator starts from the recycled blocks of three columns with all four columns: they are dropped, not misread
    it2, sol2 = A.solve(f)
    assert it2 <= it
    _close(sol2, gu.vecs(g, "sol"), 1e-6, "solution of the second solve")
    A.destroy()


@pytest.mark.parametrize("pre", ["p40", "z_p30"])
def test_richardson_and_no_krylov_match_reference(pre):
    """-hpddm_krylov_method richardson (include/HPDDM_iterative.hpp:971-993) and none (:1056-1066): the reference's solutions, real
    scalars and (round 5) K = std::complex<double>"""
    g = gu.load(pre + "_richardson_mu2")
    subs = gu.subdomains(g)
    A, d, opt = _build(g, subs)
    f = gu.vecs(g, "f")
    it, sol = A.solve(f)
    assert it == int(g["iterations_r0"][0]) == 15
    _close(sol, gu.vecs(g, "sol"), 1e-11, "Richardson")
    assert np.allclose(A.compute_residual(sol, f), g["residual_r0"], rtol=1e-9)
    A.destroy()
    g = gu.load(pre + "_none_deflated_mu2")
    subs = gu.subdomains(g)
    A, d, opt = _build(g, subs)
    it, sol = A.solve(gu.vecs(g, "f"))
    assert it == int(g["iterations_r0"][0]) == 1
    _close(sol, gu.vecs(g, "sol"), 1e-11, "one apply")
    A.destroy()


@pytest.mark.parametrize("name", ["p40_bfbcg_asm_mu3", "p40_bfbcg_asm_rhs_deflation_mu4", "z_p30_bfbcg_asm_hpd_mu3", "z_p30_bfbcg_asm_rhs_deflation_mu4", "z_p30_bfbcg_asm_shift0_mu3"])
def test_bfbcg_matches_reference(name):
    """Breakdown-free block CG (include/HPDDM_CG.hpp:342-482), plain and with -hpddm_deflation_tol on a block whose last
    right-hand side is f_0 + 2 f_1 (one direction deflated at every iteration): the reference's 31 / 32 iterations,
    residual history, solution and final residuals.  The z_ fixtures: the reference built for K = std::complex<double> on a
    Hermitian positive definite operator with complex right-hand sides -- 15 / 16 iterations (krylov_complex.hip: zbfbcg_impl)."""
    g = gu.load(name)
    subs = gu.subdomains(g)
    A, d, opt = _build(g, subs)
    f = gu.vecs(g, "f")
    it, sol, hist = A.solve(f, history=True)
    assert it == int(g["iterations_r0"][0])
    ref = g["history"]
    assert len(hist) == len(ref)
    assert np.all(np.abs(hist - ref[:, 1]) <= 1e-4 * ref[:, 1])
    _close(sol, gu.vecs(g, "sol"), 1e-7, "solution")
    assert np.allclose(A.compute_residual(sol, f), g["residual_r0"], rtol=1e-3)
    A.destroy()


@pytest.mark.parametrize("name,skip", [("p30_6ranks_bcg_asm_sym_mu2", 0), ("p40_bcg_asm_mu3", 4), ("z_p30_bcg_asm_hpd_mu3", 0), ("z_p30_6ranks_bcg_asm_hpd_mu2", 0)])
def test_bcg_matches_reference(name, skip):
    """Block CG (include/HPDDM_CG.hpp:169-337).  On the real inputs the reference's own BCG does not reach 1e-6 in 100 iterations
    (first case) or meets a rank-deficient block after 4 iterations and hands over to CG (second case, `skip` = the BCG lines
    of its log before the hand-over): what is pinned is that we do exactly the same -- iteration count, residual history,
    final residuals.  The reference tests (and prints) the residual of the LAST right-hand side against the reference norm of the
    FIRST one (include/HPDDM_CG.hpp:276): reproduced since round 5 -- the histories agree to 1e-3 where a 5 % band was needed
    while every right-hand side was tested.  The z_ fixtures: K = std::complex<double>, Hermitian positive definite operator, complex
    right-hand sides: 20 iterations on 4 and on 6 subdomains (krylov_complex.hip: zbcg_impl)."""
    g = gu.load(name)
    subs = gu.subdomains(g)
    A, d, opt = _build(g, subs)
    f = gu.vecs(g, "f")
    it, sol, hist = A.solve(f, history=True)
    assert it == int(g["iterations_r0"][0])
    ref = g["history"][skip:]
    assert len(hist) == len(ref)
    tight = 80 if name == "p30_6ranks_bcg_asm_sym_mu2" else len(ref)   # (a run that does not converge drifts in its last 20 iterations: 1e-4 ... 2e-4 in the oracle too)
    assert np.all(np.abs(hist - ref[:, 1])[:tight] <= 1e-3 * ref[:tight, 1]) and np.all(np.abs(hist - ref[:, 1]) <= 0.05 * ref[:, 1])
    if name.startswith("z_"):
        _close(sol, gu.vecs(g, "sol"), 1e-7, "solution")
    assert np.allclose(A.compute_residual(sol, f), g["residual_r0"], rtol=1e-3)
    A.destroy()


def test_full_size_properties_config2():
    """BASELINE.json configs[1] at its full size (128^3, 8 subdomains of 65^3): size-independent properties, no oracle --
    residual of the direct solves, partition of unity, linearity of the apply, agreement of the 8-right-hand-side sweep
    (MFMA forward tiles, register blocks) with eight single sweeps, GMRES iteration count and true residual."""
    import scipy.sparse as sp
    subs = generate3d(128, 8, 1, sym=True, rhs="smooth")
    A, d = hpddm.schwarz_from_subdomains(subs, options="-hpddm_operator_spd")
    A.call_numfact()
    rng = np.random.default_rng(17)
    f = [s["f"] for s in subs]
    x = A.local_solve(f)
    for sd, xs, fs in zip(subs[:2], x[:2], f[:2]):
        M = sp.csr_matrix((sd["a"], sd["ja"], sd["ia"]), shape=(sd["n"], sd["n"]))
        M = M + sp.tril(M, -1).T
        assert np.linalg.norm(M @ xs - fs) / np.linalg.norm(fs) < 1e-10
    ones = [np.ones(s["n"]) for s in subs]
    _close(A.exchange(ones), ones, 1e-14, "partition of unity")
    u = [rng.random(s["n"]) for s in subs]
    v = [rng.random(s["n"]) for s in subs]
    au, av = A.apply(u), A.apply(v)
    comb = A.apply([2.0 * a - 3.0 * b for a, b in zip(u, v)])
    _close(comb, [2.0 * a - 3.0 * b for a, b in zip(au, av)], 1e-11, "linearity of the apply")
    X = [rng.random((s["n"], 8)) for s in subs]
    Y = A.local_solve(X)
    for k in (0, 3, 7):
        single = A.local_solve([np.ascontiguousarray(xs[:, k]) for xs in X])
        _close([ys[:, k] for ys in Y], single, 1e-11, f"column {k} of the 8-right-hand-side sweep")
    it, sol = A.solve(f)
    res = A.compute_residual(sol, f)
    assert it == 26 and res[1] / res[0] <= 2e-6
    A.destroy()


@pytest.mark.parametrize("name", gu.PENALIZED_CASES)
def test_penalised_dirichlet_rows_match_reference(name):
    """same contract as tests/test_oracle_golden.py::test_penalised_dirichlet_rows_match_reference, for the HIP path"""
    g = gu.load(name)
    subs = gu.subdomains(g)
    A, d, opt = _build(g, subs)
    f = gu.vecs(g, "f")
    it, sol, hist = A.solve(f, history=True)
    ref = g["history"]
    assert it == int(g["iterations_r0"][0]) and len(hist) == len(ref)
    if opt["variant"] == "left":
        assert np.all(np.abs(hist - ref[:, 1]) <= 2e-6 * ref[:, 1])
        _close(sol, gu.vecs(g, "sol"), 1e-9, "solution")
        assert np.allclose(A.compute_residual(sol, f), g["residual_r0"], rtol=1e-5)
    else:
        assert np.all(np.abs(hist[:2] - ref[:2, 1]) <= 1e-4 * ref[:2, 1])
        assert np.allclose(A.compute_residual(sol, f)[0::2], g["residual_r0"][0::2], rtol=1e-9)
    if "residual_l1_r0" in g:   # l1 and linfty norms with penalised rows, on the reference's own solution
        rs = gu.vecs(g, "sol")
        assert np.allclose(A.compute_residual(rs, f, "l1"), g["residual_l1_r0"], rtol=1e-6)
        assert np.allclose(A.compute_residual(rs, f, "linfty"), g["residual_linfty_r0"], rtol=1e-6)
    A.destroy()


def test_custom_operator_callbacks():
    """HpddmHipSchwarzSetCustomOperator (HpddmCustomOperatorSolve, interface/hpddm_c.cpp:41-53, 227-230): the Krylov methods on an
    operator and a preconditioner given as host callbacks -- the tridiagonal matrix and the Jacobi preconditioner of
    examples/custom_operator.c:33-53 -- against a direct solve; iteration counts as the reference build of that example (6-7)."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    n, mu = 100, 2
    diag = np.arange(n) + 2.0
    T = sp.diags([-0.5 * np.ones(n - 1), diag, -0.5 * np.ones(n - 1)], [-1, 0, 1], format="csr")
    calls = {"mv": 0, "pc": 0}

    def mv(x, y):
        calls["mv"] += 1
        y[:] = T @ x

    def pc(x, y):
        calls["pc"] += 1
        y[:] = x / diag[:, None]

    eye = sp.identity(n, format="csr")
    A = hpddm.Schwarz(1)
    A.set_subdomain(0, n, eye.indptr, eye.indices, eye.data, False, [], [])
    A.initialize([np.ones(n)])
    A.set_custom_operator(mv, pc)
    rng = np.random.default_rng(11)
    b = np.asfortranarray(rng.integers(0, 10000, size=(n, mu)) / 100.0)
    exact = spla.spsolve(T.tocsc(), b)
    for opts, lo, hi in (("-hpddm_krylov_method gmres", 5, 8), ("-hpddm_krylov_method bgmres", 4, 7), ("-hpddm_krylov_method cg", 5, 9),
                         ("-hpddm_krylov_method gmres -hpddm_variant left", 5, 8)):
        A.option_parse(opts + " -hpddm_tol 1e-6")
        it, sol = A.solve([b])
        assert lo <= it <= hi, (opts, it)
        assert np.abs(sol[0] - exact).max() <= 1e-5 * np.abs(exact).max(), opts
    assert calls["mv"] > 0 and calls["pc"] > 0
    # without a preconditioner callback: identity
    A.set_custom_operator(mv, None)
    A.option_parse("-hpddm_krylov_method gmres -hpddm_variant right -hpddm_tol 1e-8 -hpddm_max_it 200")
    it, sol = A.solve([b])
    assert np.abs(sol[0] - exact).max() <= 1e-6 * np.abs(exact).max()
    A.destroy()


def test_geneo_force_uniformity_max_pads_the_short_bases():
    """-hpddm_geneo_force_uniformity max (Eigensolver::selectNu, include/HPDDM_eigensolver.hpp:121-147): the subdomains that kept fewer
    vectors than the largest basis are padded with random vectors -- uniform between the smallest and the largest entry of their own
    vectors, each made orthogonal to the first i - 1 vectors of the basis it joins as vector i (the reference's k = i - 1), not normalised --
    and the two-level operator is built on the padded bases: checked against the oracle's operator on the SAME vectors (the reference seeds
    from std::random_device: there is no run of it to compare the padding itself with)."""
    N, parts = 12, 8
    subs = generate3d(N, parts, 2, sym=True, rhs="smooth")
    A, d = hpddm.schwarz_from_subdomains(subs, options="-hpddm_operator_spd -hpddm_schwarz_coarse_correction deflated -hpddm_geneo_force_uniformity max")
    orc = Oracle(subs, correction="deflated")
    orc.multiplicity_scaling([s["d"] for s in subs])
    given = []
    for s, sd in enumerate(subs):
        i0, i1, j0, j1, k0, k1 = sd["box"]
        z, y, x = np.meshgrid(np.arange(k0, k1) / N, np.arange(j0, j1) / N, np.arange(i0, i1) / N, indexing="ij")
        Zs = np.stack([np.ones(sd["n"]), x.ravel(), y.ravel(), z.ravel()], axis=1)[:, :2 + s % 3]   # 2, 3, 4, 2, 3, 4, 2, 3 vectors (a constant alone would be padded with constants: min = max)
        Zs = np.linalg.qr(Zs)[0]   # orthonormal, as an eigensolver returns them (the reference projects with plain dot products: it relies on that)
        given.append(np.asfortranarray(Zs))
        A.set_vectors(s, Zs)
    A.build_coarse_operator()
    assert int(A.stats()["coarse_dim"]) == 4 * parts
    Z = [A.get_vectors(s) for s in range(parts)]
    for s in range(parts):
        have = given[s].shape[1]
        assert Z[s].shape == (subs[s]["n"], 4) and np.array_equal(Z[s][:, :have], given[s])
        lo, hi = given[s].min(), given[s].max()
        for i in range(have, 4):
            G = Z[s][:, :max(i - 1, 0)].T @ Z[s][:, i]
            assert np.abs(G).max(initial=0.0) <= 1e-10 * np.linalg.norm(Z[s][:, i]) * max(1.0, np.abs(Z[s][:, :i]).max())
            assert np.linalg.norm(Z[s][:, i]) > 0.1 * (hi - lo + 1e-300) and np.isfinite(Z[s][:, i]).all()
    A.call_numfact()
    orc.set_vectors(Z)
    orc.build_coarse()
    orc.numfact()
    f = orc.exchange([np.random.default_rng(3).random(sd["n"]) for sd in subs])
    _close(A.deflation(f), orc.deflation(f), 1e-9, "deflation")
    _close(A.apply(f), orc.apply(f), 1e-9, "apply")
    # the same request once more gives the same padding (the generator is seeded by the number of the subdomain)
    B, _ = hpddm.schwarz_from_subdomains(subs, options="-hpddm_operator_spd -hpddm_schwarz_coarse_correction deflated -hpddm_geneo_force_uniformity max")
    for s in range(parts):
        B.set_vectors(s, given[s])
    B.build_coarse_operator()
    assert all(np.array_equal(B.get_vectors(s), Z[s]) for s in range(parts))
    A.destroy()
    B.destroy()
