"""BASELINE.json configs[4] as SURVEY.md 8(d) C5 defines it, at sizes the oracle finishes in seconds: -Laplace(u) - k^2 u with a
first-order absorbing boundary (complex symmetric, indefinite, no volumetric damping: hpddm_amd.generate.generate_helmholtz3d),
K = std::complex<double>;
 * Schwarz::solveGEVP(A, B) for complex scalars (include/HPDDM_schwarz.hpp:665-715) with the caller's B -- the DtN slot: local Neumann
   matrix against the interface mass matrix -- and with B = scaleIntoOverlap(A): eigenvalues against ARPACK's znaupd in shift-invert
   mode, which is what the reference calls (include/HPDDM_ARPACK.hpp:84-148; scipy.sparse.linalg.eigs is that routine);
 * callNumfact(A_opt) with complex impedance matrices (ORAS, type OG) + the deflated two-level operator on the DtN space + Block GMRES
   on 8 right-hand sides: every function against the oracle built on ARPACK's vectors (the operators do not depend on the basis of
   the local spaces)."""
import numpy as np
import pytest
import scipy.sparse as sp

from hpddm_amd import hpddm
from hpddm_amd.generate import generate_helmholtz3d
from oracle import ras_oracle as ro
from oracle.ras_oracle import Oracle

pytestmark = pytest.mark.gpu


def _mat(sd, key):
    return sp.csr_matrix((sd[key], sd["ja"], sd["ia"]), shape=(sd["n"], sd["n"]))


def _bdtn(sd):
    ia, ja, a = sd["b_dtn"]
    return sp.csr_matrix((a, ja, ia), shape=(sd["n"], sd["n"]))


def _close(a, b, rtol, what):
    scale = max(np.abs(v).max() for v in b)
    err = max(np.abs(u - v).max() for u, v in zip(a, b)) / scale
    assert err <= rtol, f"{what}: relative error {err:.3e} > {rtol:.1e}"


def _operator(subs, extra=""):
    hpddm.require_device()
    A, d = hpddm.schwarz_from_subdomains(subs, options="-hpddm_schwarz_method oras " + extra, multiplicity=False)
    for s, sd in enumerate(subs):
        A.set_optimized_matrix(s, sd["n"], sd["ia"], sd["ja"], sd["a_opt"], False)
    return A, d


@pytest.mark.parametrize("user_b", [True, False])
def test_complex_gevp_against_arpack(user_b):
    """16^3 cells (k h = 3.1: the pencil does not care), 8 subdomains of 9^3: the nu eigenvalues closest to the shift, against znaupd on the
    same operator (oracle.geneo_z, itself pinned on a dense QZ in tests/test_oracle_geneo.py), and the residual of every eigenpair.
    user_b = False: B = scaleIntoOverlap(A_N), complex symmetric and indefinite -- a general pencil"""
    nu = 9
    subs = generate_helmholtz3d(16, 8, wavenumber=2.0 * np.pi * 8.0)
    A, d = _operator(subs, f"-hpddm_geneo_nu {nu} -hpddm_eigensolver_tol 1e-10")
    orc = Oracle(subs)
    orc.d = [sd["d"] for sd in subs]
    neumann = [_mat(sd, "a_neumann") for sd in subs]
    Bm = [_bdtn(sd) for sd in subs] if user_b else [orc.scale_into_overlap(s, neumann[s].astype(np.complex128)) for s in range(len(subs))]
    ref = orc.geneo_z(neumann, nu + 3, B=Bm)
    for s, sd in enumerate(subs):
        lam = A.solve_gevp(s, sd["n"], sd["ia"], sd["ja"], sd["a_neumann"], False, B=sd["b_dtn"] + (False,) if user_b else None)
        assert len(lam) == nu and np.iscomplexobj(lam)
        assert np.all(np.diff(np.abs(lam)) >= -1e-9 * np.abs(lam[-1])), "ordered by modulus"
        # every value we return is one ARPACK returned (clusters of equal modulus may be cut differently at the end of the list)
        for v in lam:
            assert np.min(np.abs(ref[s] - v)) <= 1e-6 * max(abs(v), 1e-3), (s, v, ref[s])
        X = A.get_vectors(s)
        assert X.shape == (sd["n"], nu)
        for k in range(nu):
            ax = neumann[s] @ X[:, k]
            assert np.linalg.norm(ax - lam[k] * (Bm[s] @ X[:, k])) <= 1e-7 * np.linalg.norm(ax), (s, k)
    A.destroy()


def test_dtn_threshold_keeps_the_real_parts_below_it():
    """-hpddm_geneo_threshold: Eigensolver::selectNu compares REAL parts (include/HPDDM_eigensolver.hpp:110): the DtN criterion Re(lambda) < k"""
    k = 2.0 * np.pi * 2.0
    subs = generate_helmholtz3d(16, 8, wavenumber=k)
    A, d = _operator(subs, f"-hpddm_geneo_nu 16 -hpddm_geneo_threshold {k}")
    sd = subs[3]
    lam = A.solve_gevp(3, sd["n"], sd["ia"], sd["ja"], sd["a_neumann"], False, B=sd["b_dtn"] + (False,))
    assert 1 <= len(lam) <= 16 and np.all(lam[1:].real <= k)
    assert int(A.get_option("geneo_nu")) == len(lam)
    A.destroy()


def test_oras_with_dtn_coarse_space_and_block_gmres_against_oracle():
    """configs[4] in small: 2 x 2 x 2 subdomains of a 20^3 grid, k = 2 pi 2 (k h = 0.63), ORAS with impedance matrices, DtN coarse space
    from solveGEVP(A_N, B_Gamma) (nu = 6 = two whole triples of these cubic subdomains: inside a cluster the two eigensolvers differ by a
    rotation, which the operators do not see), Block GMRES on 8 right-hand sides"""
    k, nu, mu = 2.0 * np.pi * 2.0, 6, 8
    subs = generate_helmholtz3d(20, 8, wavenumber=k)
    A, d = _operator(subs, f"-hpddm_geneo_nu {nu} -hpddm_eigensolver_tol 1e-11 -hpddm_schwarz_coarse_correction deflated -hpddm_krylov_method bgmres -hpddm_gmres_restart 30 -hpddm_max_it 200")
    orc = Oracle(subs, correction="deflated", method="oras")
    orc.d = [sd["d"] for sd in subs]
    lam_ref = orc.geneo_z([_mat(sd, "a_neumann") for sd in subs], nu, B=[_bdtn(sd) for sd in subs])
    for s, sd in enumerate(subs):
        lam = A.solve_gevp(s, sd["n"], sd["ia"], sd["ja"], sd["a_neumann"], False, B=sd["b_dtn"] + (False,))
        assert len(lam) == nu and np.all(np.abs(np.abs(lam) - np.abs(lam_ref[s])) <= 1e-6 * np.abs(lam_ref[s])), (s, lam, lam_ref[s])
    A.build_coarse_operator()
    A.call_numfact()
    orc.build_coarse(lapacktr=False)
    orc.numfact([_mat(sd, "a_opt") for sd in subs])
    rs = np.random.RandomState(42)   # mt19937(seed = 42), uniform(0, 1) re / im
    f = orc.exchange([rs.random_sample((sd["n"], mu)) + 1j * rs.random_sample((sd["n"], mu)) for sd in subs])
    _close(A.exchange(f), orc.exchange(f), 1e-14, "exchange")
    _close(A.gmv(f), orc.gmv(f), 1e-13, "GMV")
    _close(A.local_solve(f), orc.local_solve(f), 1e-9, "Solver::solve on the impedance matrices")
    _close(A.deflation(f), orc.deflation(f), 1e-6, "deflation on the DtN space")
    _close(A.apply(f), orc.apply(f), 1e-6, "two-level ORAS apply")
    it, sol = A.solve(f)
    it_o, sol_o, _ = ro.bgmres(orc, f, restart=30, max_it=200)
    assert abs(it - it_o) <= 1 and it < 60, (it, it_o)
    res = A.compute_residual(sol, f).reshape(mu, 2)
    assert np.all(res[:, 1] <= 2e-6 * res[:, 0]), res
    _close(sol, sol_o, 1e-4, "solution")
    A.destroy()
