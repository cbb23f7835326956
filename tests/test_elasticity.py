"""3-D linear elasticity workload (BASELINE.json configs[3]: block-3 CSR, RAS + GenEO): the generator against a global
assembly, the oracle on it (CPU), and the HIP path against the oracle (GPU)."""
import numpy as np
import pytest
import scipy.sparse as sp

from hpddm_amd.generate import _q1_elasticity_stiffness, generate_elasticity3d
from oracle.ras_oracle import Oracle, csr_full


def _neumann(subs):
    out = []
    for sd in subs:
        sn = dict(sd)
        sn["ia"], sn["ja"], sn["a"] = sd["ia_neumann"], sd["ja_neumann"], sd["a_neumann"]
        out.append(csr_full(sn))
    return out


def test_element_matrix_has_six_rigid_body_modes():
    ev = np.linalg.eigvalsh(_q1_elasticity_stiffness(0.25, 1.0, 0.3))
    assert np.all(np.abs(ev[:6]) < 1e-12) and ev[6] > 1e-3


@pytest.mark.parametrize("overlap", [1, 2])
def test_subdomain_matrices_are_restrictions_of_the_global_one(overlap):
    N, parts = 9, 8
    g = generate_elasticity3d(N, 1, overlap=0, sym=False, normalize=False)[0]
    G = sp.csr_matrix((g["a"], g["ja"], g["ia"]), shape=(g["n"], g["n"]))
    assert abs(G - G.T).max() < 1e-14
    subs = generate_elasticity3d(N, parts, overlap=overlap, sym=True, neumann=True)
    pou = np.zeros(g["n"])
    for sd in subs:
        i0, i1, j0, j1, k0, k1 = sd["box"]
        nodes = np.arange(N ** 3).reshape(N, N, N)[k0:k1, j0:j1, i0:i1].reshape(-1)
        dof = (3 * nodes[:, None] + np.arange(3)).reshape(-1)
        assert abs(csr_full(sd) - G[dof][:, dof]).max() < 1e-14
        pou[dof] += sd["d"]
        # unassembled local matrix: positive semi-definite, and R A R^T - A_N only touches the artificial interface
        AN = _neumann([sd])[0]
        assert np.linalg.eigvalsh(AN.toarray())[0] > -1e-12
    assert np.allclose(pou, 1.0)  # sum_i R_i^T D_i R_i = I


def test_oracle_two_level_geneo_on_elasticity():
    subs = generate_elasticity3d(8, 8, overlap=1, sym=True, neumann=True)
    orc = Oracle(subs)
    orc.d = [s["d"] for s in subs]
    orc.numfact()
    f = [s["f"] for s in subs]
    it1, sol1, _ = orc.gmres(f, max_it=200)
    lam = orc.geneo(_neumann(subs), 8)
    assert all(np.all(np.abs(l[:6]) < 1e-6) for i, l in enumerate(lam) if subs[i]["box"][0] > 0)  # floating subdomains: rigid-body modes
    orc.correction = "deflated"
    orc.build_coarse()
    it2, sol2, _ = orc.gmres(f, max_it=200)
    assert it2 < it1
    res = orc.compute_residual(sol2, f)
    assert res[1] / res[0] < 1e-5


@pytest.mark.gpu
def test_hip_against_oracle_on_elasticity():
    from hpddm_amd import hpddm
    nu = 8
    subs = generate_elasticity3d(10, 8, overlap=1, sym=True, neumann=True)
    A, d = hpddm.schwarz_from_subdomains(subs, options=f"-hpddm_operator_spd -hpddm_geneo_nu {nu} -hpddm_eigensolver_tol 1e-9", multiplicity=False)
    orc = Oracle(subs)
    orc.d = [s["d"] for s in subs]
    A.call_numfact()
    orc.numfact()

    def close(a, b, tol, what):
        scale = max(np.abs(v).max() for v in b)
        err = max(np.abs(u - v).max() for u, v in zip(a, b)) / scale
        assert err < tol, (what, err)

    rng = np.random.default_rng(3)
    x = [rng.random((s["n"], 2)) for s in subs]
    close(A.exchange(x), orc.exchange(x), 1e-14, "exchange")
    close(A.gmv(x), orc.gmv(x), 1e-12, "gmv")
    f = orc.exchange(x)
    close(A.apply(f), orc.apply(f), 1e-9, "one-level apply")
    fr = [s["f"] for s in subs]
    it, sol = A.solve(fr)
    it_o, sol_o, _ = orc.gmres(fr)
    assert it == it_o, (it, it_o)
    close(sol, sol_o, 1e-6, "solution")
    lam_ref = orc.geneo(_neumann(subs), nu)
    for s, sd in enumerate(subs):
        lam = A.solve_gevp(s, sd["n"], sd["ia_neumann"], sd["ja_neumann"], sd["a_neumann"], sd["sym"])
        assert len(lam) == nu
        assert np.all(np.abs(lam - lam_ref[s]) <= 1e-6 * np.maximum(np.abs(lam_ref[s]), 1e-2)), (s, lam, lam_ref[s])
    A.build_coarse_operator()
    orc.build_coarse()
    A.option_parse("-hpddm_schwarz_coarse_correction deflated")
    orc.correction = "deflated"
    close(A.apply(f), orc.apply(f), 1e-6, "two-level apply")
    it2, sol2 = A.solve(fr)
    it2_o, _, _ = orc.gmres(fr)
    assert abs(it2 - it2_o) <= 1 and it2 < it
    res = A.compute_residual(sol2, fr)
    assert res[1] / res[0] < 1e-5
    A.destroy()
