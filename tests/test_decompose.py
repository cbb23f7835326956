"""hpddm_amd/decompose.py (the counterpart of the reference's examples/generateFromFile.cpp): an algebraically decomposed global matrix
through the oracle on CPU -- partition of unity, consistency of the shared lists, RAS-preconditioned GMRES against the global direct
solution -- and through the library on the GPU."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spl

from hpddm_amd.decompose import decompose, gather, strip_partition


def _global_problem(nx=24, ny=18):
    """5-point Laplacian with a convective term (non-symmetric), like a user matrix would be"""
    n = nx * ny
    A = sp.lil_matrix((n, n))
    for j in range(ny):
        for i in range(nx):
            k = i + nx * j
            A[k, k] = 4.2
            if i > 0:
                A[k, k - 1] = -1.3
            if i < nx - 1:
                A[k, k + 1] = -0.7
            if j > 0:
                A[k, k - nx] = -1.0
            if j < ny - 1:
                A[k, k + nx] = -1.0
    b = np.sin(0.1 * np.arange(n)) + 1.0
    return A.tocsr(), b


@pytest.mark.parametrize("parts,overlap", [(4, 1), (6, 2), (5, 3)])
def test_decomposition_through_the_oracle(parts, overlap):
    from oracle.ras_oracle import Oracle
    A, b = _global_problem()
    n = A.shape[0]
    subs = decompose(A, parts, overlap, rhs=b)
    part = strip_partition(A, parts)
    assert all(np.all(np.diff(sd["idx"]) > 0) for sd in subs)
    for p, sd in enumerate(subs):       # own part inside, weights as in generateFromFile.cpp:114-118
        own = part[sd["idx"]] == p
        assert np.all(sd["d"][own] == 1.0) and np.all((sd["d"][~own] >= 0.0) & (sd["d"][~own] < 1.0))
        for q, conn in zip(sd["neighbors"], sd["connectivity"]):   # both sides list the same global unknowns in the same order
            back = subs[q]["connectivity"][list(subs[q]["neighbors"]).index(p)]
            assert np.array_equal(sd["idx"][conn], subs[q]["idx"][back])
    orc = Oracle(subs)
    d = orc.multiplicity_scaling([sd["d"] for sd in subs])
    assert np.allclose(sum(np.bincount(sd["idx"], weights=dd, minlength=n) for sd, dd in zip(subs, d)), 1.0)   # partition of unity
    orc.numfact()
    f = [sd["f"] for sd in subs]
    x_ref = spl.spsolve(A.tocsc(), b)
    assert np.allclose(gather(subs, orc.gmv([x_ref[sd["idx"]] for sd in subs]), n), b)      # GMV of the duplicated vector = global product
    it, sol, _ = orc.gmres(f, tol=1e-10, max_it=200)
    assert it < 60
    assert np.abs(gather(subs, sol, n) - x_ref).max() <= 1e-7 * np.abs(x_ref).max()


@pytest.mark.gpu
def test_decomposition_on_the_device():
    from hpddm_amd import hpddm
    A, b = _global_problem(40, 30)
    n = A.shape[0]
    subs = decompose(A, 8, 2, rhs=b)
    S, d = hpddm.schwarz_from_subdomains(subs, options="-hpddm_tol 1e-10 -hpddm_max_it 200")
    S.call_numfact()
    it, sol = S.solve([sd["f"] for sd in subs])
    x_ref = spl.spsolve(A.tocsc(), b)
    assert it < 80 and np.abs(gather(subs, sol, n) - x_ref).max() <= 1e-7 * np.abs(x_ref).max()
    S.destroy()


@pytest.mark.gpu
def test_schwarz_from_file_example():
    """examples/schwarz_from_file.py (schwarzFromFile.cpp): a matrix dumped by the reference, split into 3 overlapping subdomains"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "examples", "schwarz_from_file.py"), "-matrix_filename=" + os.path.join(root, "tests", "golden", "dump", "out_0_4.txt"),
                          "--subdomains", "3", "-overlap", "2", "-hpddm_tol", "1e-8"], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "--- residual" in res.stdout, res.stdout + res.stderr
