"""The CPU restatement (oracle/ras_oracle.py) against the golden vectors dumped from the compiled reference.

This is what pins the oracle: every function of the hot path, on every fixture, plus the 45-iteration run of
BASELINE.json's config 1.  CPU only.
"""
import numpy as np
import pytest

import golden_util as gu
from hpddm_amd.generate import generate2d
from oracle.ras_oracle import Oracle


def _setup(g, subs):
    from oracle.ras_oracle import csr_full
    opt = gu.options(g)
    orc = Oracle(subs, correction=opt["correction"], method=opt["method"])
    orc.multiplicity_scaling([s["d"] for s in subs])
    if "a_opt_r0" in g:   # callNumfact(A_opt): ORAS / SORAS with an optimised local matrix
        orc.numfact([csr_full(t) for t in gu.optimized_matrices(g, subs)])
    else:
        orc.numfact()
    if opt["correction"]:
        orc.set_vectors(gu.deflation_vectors(g, subs))
        orc.build_coarse()
    return orc, opt


def _close(a, b, rtol, what):
    scale = max(np.abs(np.concatenate([np.ravel(x) for x in b])).max(), 1e-300)
    err = max(np.abs(np.ravel(x) - np.ravel(y)).max() for x, y in zip(a, b)) / scale
    assert err <= rtol, f"{what}: relative error {err:.3e} > {rtol:.1e}"


@pytest.mark.parametrize("name", gu.SMALL_CASES + gu.OPTIMIZED_CASES + gu.PENALIZED_CASES + gu.MULTI_VECTOR_CASES + gu.COMPLEX_CASES
                         + gu.COMPLEX_BGMRES_CASES)
def test_functions_match_reference(name):
    g = gu.load(name)
    subs = gu.subdomains(g)
    orc, opt = _setup(g, subs)
    P, mu = int(g["ranks"]), int(g["mu"])
    # Subdomain::initialize + multiplicityScaling
    for r in range(P):
        assert [q for q, _ in orc.map[r]] == list(g[f"neighbors_r{r}"])
        for k, (_, idx) in enumerate(orc.map[r]):
            assert np.array_equal(idx, g[f"map_{k}_r{r}"])
        assert np.abs(orc.d[r] - g[f"d_r{r}"]).max() <= 1e-15
    f = gu.vecs(g, "f")
    _close(orc.exchange(f), gu.vecs(g, "exchange_out"), 1e-14, "exchange")
    _close(orc.gmv(f), gu.vecs(g, "gmv_out"), 1e-13, "GMV")
    _close(orc.local_solve(f), gu.vecs(g, "solve_out"), 1e-11, "Solver::solve")
    loose = 1e-8 if "nu3" in name else 1e-10   # three smooth vectors per subdomain: the coarse matrix is ill-conditioned
    _close(orc.apply(f), gu.vecs(g, "apply_out"), loose, "apply")
    if opt["correction"]:
        _close(orc.deflation(f), gu.vecs(g, "deflation_out"), loose, "deflation")


@pytest.mark.parametrize("name", gu.SMALL_CASES + gu.OPTIMIZED_CASES + gu.MULTI_VECTOR_CASES + gu.COMPLEX_CASES)
def test_gmres_matches_reference(name):
    g = gu.load(name)
    subs = gu.subdomains(g)
    orc, opt = _setup(g, subs)
    it, sol, hist = orc.gmres(gu.vecs(g, "f"), tol=opt["tol"], max_it=opt["max_it"], restart=opt["restart"], variant=opt["variant"], ortho=opt["ortho"])
    assert it == int(g["iterations_r0"][0])
    ref_hist = g["history"]
    assert len(hist) == len(ref_hist)
    for (j, beta, nrm), row in zip(hist, ref_hist):
        assert j == int(row[0])
        assert abs(beta - row[1]) <= (2e-3 if "nu3" in name else 2e-6) * row[1] + 1e-300   # the log prints 7 significant digits
    _close(sol, gu.vecs(g, "sol"), 1e-6 if "nu3" in name else 1e-8, "solution")
    res = orc.compute_residual(sol, gu.vecs(g, "f"))
    assert np.allclose(res, g["residual_r0"], rtol=1e-5)
    if "residual_l1_r0" in g:   # the other norms of Schwarz::computeResidual, on the reference's own solution
        rs, rf = gu.vecs(g, "sol"), gu.vecs(g, "f")
        assert np.allclose(orc.compute_residual(rs, rf, "l1"), g["residual_l1_r0"], rtol=1e-6)
        assert np.allclose(orc.compute_residual(rs, rf, "linfty"), g["residual_linfty_r0"], rtol=1e-6)


def test_config1_45_iterations():
    """BASELINE.json config 1: 2-D Poisson 200x200, 4 subdomains, one-level RAS -> 45 iterations"""
    g = gu.load("c1_p200_onelevel")
    subs = generate2d(200, 200, 4)
    orc = Oracle(subs)
    orc.multiplicity_scaling([s["d"] for s in subs])
    orc.numfact()
    f = [s["f"] for s in subs]
    _close(orc.apply(f), [g[f"apply_out_r{r}"] for r in range(4)], 1e-10, "apply")
    it, sol, hist = orc.gmres(f)
    assert it == 45 == int(g["iterations_r0"][0])
    for (j, beta, nrm), row in zip(hist, g["history"]):
        assert abs(beta - row[1]) <= 1e-4 * row[1]  # 45 iterations and a restart amplify round-off differences
    _close(sol, [g[f"sol_r{r}"] for r in range(4)], 1e-8, "solution")
    assert np.allclose(orc.compute_residual(sol, f), g["residual_r0"], rtol=1e-5)


@pytest.mark.parametrize("name", gu.PENALIZED_CASES)
def test_penalised_dirichlet_rows_match_reference(name):
    """FreeFEM-style boundary conditions (diagonal HPDDM_PEN = 1e30 on every 11th grid point, right-hand side HPDDM_PEN * g):
    Schwarz::start, initializeNorm and computeResidual treat those rows apart (include/HPDDM_schwarz.hpp:496-514, 761-803,
    include/HPDDM_iterative.hpp:441-471).  The left-preconditioned run converges in 8 iterations and is pinned to round-off;
    the right-preconditioned one does not converge in the reference either (100 iterations) and amplifies round-off, so
    only its count, its initial norm (where the penalised entries count divided by HPDDM_PEN) and its first two residuals are
    compared."""
    g = gu.load(name)
    subs = gu.subdomains(g)
    orc, opt = _setup(g, subs)
    bc = orc.boundary_conditions()
    assert sum(int((b != 0).sum()) for b in bc) > 0 and all(np.all((b == 0) | (b == 1.0e30)) for b in bc)
    f = gu.vecs(g, "f")
    it, sol, hist = orc.gmres(f, tol=opt["tol"], max_it=opt["max_it"], restart=opt["restart"], variant=opt["variant"], ortho=opt["ortho"])
    ref = g["history"]
    assert it == int(g["iterations_r0"][0]) and len(hist) == len(ref)
    assert abs(hist[0][2] - ref[0, 2]) <= 2e-6 * ref[0, 2]
    if opt["variant"] == "left":
        for (j, beta, nrm), row in zip(hist, ref):
            assert abs(beta - row[1]) <= 2e-6 * row[1]
        _close(sol, gu.vecs(g, "sol"), 1e-10, "solution")
        assert np.allclose(orc.compute_residual(sol, f), g["residual_r0"], rtol=1e-6)
    else:
        for (j, beta, nrm), row in list(zip(hist, ref))[:2]:
            assert abs(beta - row[1]) <= 1e-4 * row[1]
        assert np.allclose(orc.compute_residual(sol, f)[0::2], g["residual_r0"][0::2], rtol=1e-9)   # ||f|| with the penalised entries
    if "residual_l1_r0" in g:   # l1 and linfty norms with penalised rows, on the reference's own solution
        rs = gu.vecs(g, "sol")
        assert np.allclose(orc.compute_residual(rs, f, "l1"), g["residual_l1_r0"], rtol=1e-6)
        assert np.allclose(orc.compute_residual(rs, f, "linfty"), g["residual_linfty_r0"], rtol=1e-6)


@pytest.mark.parametrize("name,method,tol_hist", [
    ("p40_cg_asm", "cg", 2e-6), ("p40_bgmres_mu4", "bgmres", 2e-6), ("p40_bgmres_deflated_mu2", "bgmres", 2e-4),
    ("p30_6ranks_bgmres_left_mu3", "bgmres", 2e-6), ("p40_fbgmres_mu3", "bgmres", 2e-4),
    ("p40_bgmres_rhs_deflation_mu4", "bgmres", 2e-6), ("p40_bgmres_rhs_deflation_restart_mu4", "bgmres", 5e-5),
    ("z_p30_6ranks_bgmres_mu3_balanced", "bgmres", 2e-6), ("z_p30_bgmres_mu8", "bgmres", 2e-6), ("z_p30_bgmres_rhs_deflation_mu4", "bgmres", 2e-6), ("z_p30_fbgmres_mu3", "bgmres", 2e-4),
    ("p40_bgmres_mgs_qrmgs_mu3", "bgmres", 1e-5), ("p40_bgmres_qrcgs_mu3", "bgmres", 1e-5),
    ("p40_bfbcg_asm_mu3", "bfbcg", 1e-5), ("p40_bfbcg_asm_rhs_deflation_mu4", "bfbcg", 1e-5),
    ("p30_6ranks_bcg_asm_sym_mu2", "bcg", 5e-4), ("p40_bcg_asm_mu3", "bcg", 2e-6),
    # K = std::complex<double> on a Hermitian positive definite operator: the reference's CG / BCG / BFBCG converge (17 / 20 / 15 / 16 / 20 iterations)
    ("z_p30_cg_asm_hpd_mu3", "cg", 1e-5), ("z_p30_bcg_asm_hpd_mu3", "bcg", 1e-4), ("z_p30_bfbcg_asm_hpd_mu3", "bfbcg", 1e-4),
    ("z_p30_bfbcg_asm_rhs_deflation_mu4", "bfbcg", 1e-4), ("z_p30_6ranks_bcg_asm_hpd_mu2", "bcg", 1e-4),
    # the same operator without its shift (the plain Laplacian): BFBCG 34 iterations
    ("z_p30_bfbcg_asm_shift0_mu3", "bfbcg", 1e-4)])
def test_other_krylov_methods_match_reference(name, method, tol_hist):
    """CG, Block CG and Block GMRES restated in numpy (oracle/ras_oracle.py: cg, bcg, bgmres) against the runs of the compiled
    reference: iteration counts, residual histories, final residuals.  (BCG tests and prints the residual of the LAST right-hand
    side against the reference norm of the FIRST one, include/HPDDM_CG.hpp:276 -- reproduced; the 100-iteration run that does not
    converge drifts by 2e-4 at its end.  In
    the second BCG fixture the reference meets a rank-deficient block after 4 iterations and hands over to CG: so do we.
    The BFBCG fixtures (breakdown-free block CG, include/HPDDM_CG.hpp:342-482) stop at 1e-4, before CG's round-off drift.
    The two rhs_deflation fixtures have a last right-hand side f_0 + 2 f_1 and -hpddm_deflation_tol set: one column is
    deflated at every restart, include/HPDDM_GMRES.hpp:201-205.)"""
    from oracle import ras_oracle as ro
    g = gu.load(name)
    subs = gu.subdomains(g)
    orc, opt = _setup(g, subs)
    f = gu.vecs(g, "f")
    ref = g["history"]
    if method == "cg":
        it, sol, hist = ro.cg(orc, f, tol=opt["tol"], max_it=opt["max_it"])
    elif method == "bgmres":
        it, sol, hist = ro.bgmres(orc, f, tol=opt["tol"], max_it=opt["max_it"], restart=opt["restart"], variant=opt["variant"],
                                  deflation_tol=opt["deflation_tol"], ortho=opt["ortho"], qr=opt["qr"])
    elif method == "bfbcg":
        it, sol, hist = ro.bfbcg(orc, f, tol=opt["tol"], max_it=opt["max_it"], deflation_tol=opt["deflation_tol"])
    else:
        it, sol, hist, handed_over = ro.bcg(orc, f, tol=opt["tol"], max_it=opt["max_it"])
        assert handed_over == (name == "p40_bcg_asm_mu3")
        ref = ref[len(ref) - len(hist):]
    assert it == int(g["iterations_r0"][0]) and len(hist) == len(ref)
    for (j, beta, nrm), row in zip(hist, ref):
        assert abs(beta - row[1]) <= tol_hist * row[1]
    assert np.allclose(orc.compute_residual(sol, f), g["residual_r0"], rtol=1e-3)


def _gcrodr_block(orc, f, opt, recycle, state, same_system):
    """the non-block method one right-hand side at a time (the reference runs them in lock-step: same iterates); history =
    per iteration, the largest residual among the right-hand sides still iterating (checkConvergence)"""
    from oracle import ras_oracle as ro
    mu = 1 if f[0].ndim == 1 else f[0].shape[1]
    runs = []
    for nu in range(mu):
        fn = [v if v.ndim == 1 else v[:, nu] for v in f]
        runs.append(ro.gcrodr(orc, fn, tol=opt["tol"], max_it=opt["max_it"], restart=opt["restart"], recycle=recycle, variant=opt["variant"],
                              ortho=opt["ortho"], state=None if state is None else state[nu], same_system=same_system,
                              target=opt["recycle_target"]))
    it = max(r[0] for r in runs)
    hist = []
    for j in range(it):   # checkConvergence prints the residual of the first right-hand side unless one still iterating has a larger one
        beta = runs[0][2][min(j, len(runs[0][2]) - 1)][1]
        hist.append(max([beta] + [r[2][j][1] for r in runs if len(r[2]) > j + 1]))
    sol = [np.stack([r[1][s] for r in runs], axis=1) if mu > 1 else runs[0][1][s] for s in range(orc.P)]
    # (the lock-step run of the reference keeps iterating -- and printing -- a first right-hand side that has converged while another one
    # has not: past that point its printed residual cannot come from a replay that stops every right-hand side at its own convergence;
    # the solution does not depend on it, updateSol uses the dimension each right-hand side converged at)
    _gcrodr_block.printed = len(runs[0][2])
    return it, sol, hist, [r[3] for r in runs]


@pytest.mark.parametrize("name,recycle,same", [("p40_gcrodr_two_solves", 4, 0), ("p40_gcrodr_same_system", 4, 1),
                                               ("p30_6ranks_gcrodr_left_deflated_mu2", 3, 0), ("p40_gcrodr_target_lm", 4, 0),
                                               ("p40_gcrodr_cycle_end", 4, 0),
                                               # K = std::complex<double> (the reference's harness built for complex scalars): 33 + 33, 19 + 18 (two
                                               # right-hand sides), 17 + 15 (left, MGS) and 22 + 22 iterations (LM, frozen subspace)
                                               ("z_p30_gcrodr_two_solves", 4, 0), ("z_p30_gcrodr_mu2", 3, 0), ("z_p30_gcrodr_left_mgs", 4, 0),
                                               ("z_p30_gcrodr_target_lm_same_system", 3, 1)])
def test_gcrodr_matches_reference(name, recycle, same):
    """GCRO-DR (include/HPDDM_GCRODR.hpp:34-443) on two successive solves: the first builds the recycled subspace from the
    harmonic Ritz vectors of its first cycle and updates it at every restart, the second starts from it (19 then 15
    iterations instead of GMRES(10)'s 24).  With -hpddm_recycle_same_system the reference increments the option after the
    first solve, which freezes the subspace during the second one."""
    g = gu.load(name)
    subs = gu.subdomains(g)
    orc, opt = _setup(g, subs)
    f, f2 = gu.vecs(g, "f"), gu.vecs(g, "f2")
    it, sol, hist, state = _gcrodr_block(orc, f, opt, recycle, None, same)
    n1 = _gcrodr_block.printed
    it2, sol2, hist2, _ = _gcrodr_block(orc, f2, opt, recycle, state, 2 * same)
    n2 = _gcrodr_block.printed
    assert it == int(g["iterations_r0"][0]) and it2 == int(g["iterations2_r0"][0])
    ref = g["history"][:, 1]
    assert len(ref) == it + it2
    assert np.allclose(hist[:n1], ref[:n1], rtol=1e-4) and np.allclose(hist2[:n2], ref[it:it + n2], rtol=1e-4)
    assert n1 >= it - 3 and n2 >= it2 - 3      # (the fixtures' right-hand sides converge within a few iterations of each other)
    _close(sol, gu.vecs(g, "sol"), 1e-9, "solution")
    _close(sol2, gu.vecs(g, "sol2"), 1e-9, "second solution")


@pytest.mark.parametrize("pre", ["p40", "z_p30"])
def test_richardson_and_no_krylov_match_reference(pre):
    """-hpddm_krylov_method richardson (15 damped iterations, no convergence test) and none (one preconditioner apply): the
    reference's solutions.  (With `none` the reference hands its right-hand side to the two-level apply as work space, so the
    residual it prints afterwards is not that of the system: only the solution is compared.)"""
    from oracle import ras_oracle as ro
    g = gu.load(pre + "_richardson_mu2")
    subs = gu.subdomains(g)
    orc, opt = _setup(g, subs)
    f = gu.vecs(g, "f")
    it, sol = ro.richardson(orc, f, opt["max_it"], 0.7)
    assert it == int(g["iterations_r0"][0]) == 15
    _close(sol, gu.vecs(g, "sol"), 1e-12, "Richardson")
    assert np.allclose(orc.compute_residual(sol, f), g["residual_r0"], rtol=1e-10)
    g = gu.load(pre + "_none_deflated_mu2")
    subs = gu.subdomains(g)
    orc, opt = _setup(g, subs)
    it, sol = ro.no_krylov(orc, gu.vecs(g, "f"))
    assert it == int(g["iterations_r0"][0]) == 1
    _close(sol, gu.vecs(g, "sol"), 1e-12, "one apply")


@pytest.mark.parametrize("name,recycle", [("p40_bgcrodr_two_solves_mu2", 3), ("p30_6ranks_bgcrodr_left_deflated_mu3", 2),
                                          # K = std::complex<double>: 36 + 32 and 19 + 18 iterations of the reference built for complex scalars
                                          ("z_p30_bgcrodr_two_solves_mu2", 3), ("z_p30_6ranks_bgcrodr_left_mu3", 2)])
def test_block_gcrodr_matches_reference(name, recycle):
    """Block GCRO-DR (include/HPDDM_GCRODR.hpp:445-905), two successive solves: 18 then 13 iterations for two right-hand sides.
    In the first fixture a complex pair of harmonic Ritz values is cut by the selection after the first cycle, and the first
    solve ends on the last step of a cycle (the reference then builds the recycled space with an un-normalised last block):
    both conventions are reproduced."""
    from oracle import ras_oracle as ro
    g = gu.load(name)
    subs = gu.subdomains(g)
    orc, opt = _setup(g, subs)
    f, f2 = gu.vecs(g, "f"), gu.vecs(g, "f2")
    it, sol, hist, state = ro.bgcrodr(orc, f, tol=opt["tol"], max_it=opt["max_it"], restart=opt["restart"], recycle=recycle, variant=opt["variant"])
    it2, sol2, hist2, _ = ro.bgcrodr(orc, f2, tol=opt["tol"], max_it=opt["max_it"], restart=opt["restart"], recycle=recycle, variant=opt["variant"], state=state)
    assert it == int(g["iterations_r0"][0]) and it2 == int(g["iterations2_r0"][0])
    ref = g["history"][:, 1]
    assert len(ref) == it + it2
    assert np.allclose([h[1] for h in hist], ref[:it], rtol=1e-4) and np.allclose([h[1] for h in hist2], ref[it:], rtol=1e-4)
    _close(sol, gu.vecs(g, "sol"), 1e-9, "solution")
    _close(sol2, gu.vecs(g, "sol2"), 1e-9, "second solution")


@pytest.mark.parametrize("name", ["p40_bgcrodr_rhs_deflation_mu4", "z_p30_bgcrodr_rhs_deflation_mu4"])
def test_block_gcrodr_with_rhs_deflation_matches_reference(name):
    """Block GCRO-DR with -hpddm_deflation_tol (include/HPDDM_GCRODR.hpp:545-600): the last of the four right-hand sides is f_0 + 2 f_1,
    the cycles run on blocks of three columns, the recycled space and its eigenproblems with them; 17 iterations (real), 32 (complex)
    of the compiled reference, histories and solutions."""
    from oracle import ras_oracle as ro
    g = gu.load(name)
    subs = gu.subdomains(g)
    orc, opt = _setup(g, subs)
    f = gu.vecs(g, "f")
    it, sol, hist, _ = ro.bgcrodr(orc, f, tol=opt["tol"], max_it=opt["max_it"], restart=opt["restart"], recycle=2, variant=opt["variant"], deflation_tol=1e-6)
    ref = g["history"][:, 1]
    assert it == int(g["iterations_r0"][0]) == len(ref)
    assert np.allclose([h[1] for h in hist], ref, rtol=1e-4)
    _close(sol, gu.vecs(g, "sol"), 1e-8, "solution")
    assert np.allclose(orc.compute_residual(sol, f), g["residual_r0"], rtol=1e-5)
