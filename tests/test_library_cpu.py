"""CPU-side checks of the product library: the C ABI exports what include/hpddm_hip.h declares, the host analysis /
factorisation is correct (validated by replaying the multifrontal solve in numpy on the exported panels), and the
compute entry points fail loudly without a GPU instead of falling back to anything."""
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp

from hpddm_amd import _lib, hpddm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported():
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "hpddm_hip.h")).read()
    declared = set(re.findall(r"\b(HpddmHip\w+)\s*\(", header))
    declared -= {"HpddmHipSubdomain", "HpddmHipSchwarz"}
    assert len(declared) >= 35
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in hpddm_hip.h but not exported"
    assert set(_lib.DECLARED_SYMBOLS) == declared, declared ^ set(_lib.DECLARED_SYMBOLS)


def _poisson3d(N):
    I = sp.identity(N)
    T = sp.diags([-1, 2, -1], [-1, 0, 1], shape=(N, N))
    return (sp.kron(sp.kron(T, I), I) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(I, I), T)).tocsr()


def _leaf_blob(pool, off, w, ldw, nb, nnzr, nnzc, cs):
    """sections of a condensed leaf's blob (leaf_blob_layout, hpddm_amd/csrc/factor.hpp); ldw in doubles"""
    raw = pool.view(np.uint8)[8 * off:]
    dt = np.complex128 if cs == 2 else np.float64
    o_srval = w * ldw * 8
    o_scval = o_srval + nnzr * cs * 8
    o_scrow = o_scval + nnzc * cs * 8
    o_srptr = o_scrow + nnzc * 4
    o_scptr = o_srptr + (nb + 1) * 2
    o_srcol = o_scptr + (w + 1) * 2
    WT = raw[:o_srval].view(dt).reshape(w, ldw // cs)[:, :w]
    return dict(W=WT.T, srval=raw[o_srval:o_scval].view(dt), scval=raw[o_scval:o_scrow].view(dt), scrow=raw[o_scrow:o_srptr].view(np.int32),
                srptr=raw[o_srptr:o_scptr].view(np.uint16), scptr=raw[o_scptr:o_srcol].view(np.uint16), srcol=raw[o_srcol:o_srcol + 2 * nnzr].view(np.uint16))


def _replay(S, b, use_leaves=True, compact=False):
    """numpy replay of the level-scheduled multifrontal solve on the exported solve-ready panels (factor.hpp); complex factors:
    the panels are (re, im) pairs, L D L^T with plain transposes (complex symmetric) or LU.  The updates travel as in the sweeps
    of the library: every supernode writes its update to positions rel[.] of a slot row of its parent, the parent sums its slot rows
    (dense); a condensed leaf goes through W = inv(A_JJ) and the sparse couplings of its blob (use_leaves=False: through its panel).
    compact=True: the hand-over of the 16-column engine instead -- one entry per (position of a front, child that reaches it), runs
    named by cptr, written through crel."""
    e = {k: S.export(k) for k in ("perm", "blk_ptr", "ldw", "f_off", "row_ptr", "rows", "height", "u_off", "rel", "nchild", "s_off", "ps_off", "tgs", "lb_off", "lb_nnzr", "lb_nnzc", "c_off", "cptr", "crel", "cs_off", "pcs_off")}
    pool = S.export("leaf_pool")
    kind = S.info()["kind"]
    cplx = bool(getattr(S, "complex", False))
    dt = np.complex128 if cplx else np.float64
    cs = 2 if cplx else 1
    F = S.export("F")
    G = S.export("G") if kind == 2 else F
    dinv = S.export("dinv") if kind == 1 else None
    if cplx:
        F, G = F.view(np.complex128), G.view(np.complex128)
        dinv = dinv.view(np.complex128) if dinv is not None else None
    n, blk, rp = len(e["perm"]), e["blk_ptr"], e["row_ptr"]
    nblk = len(blk) - 1
    hh = np.diff(blk) + np.diff(rp)
    slots = np.zeros(max(1, int((e["nchild"] * hh).sum())), dtype=dt)
    assert e["s_off"][-1] + e["nchild"][-1] * hh[-1] == (e["nchild"] * hh).sum()
    y, x = np.zeros(n, dtype=dt), np.zeros(n, dtype=dt)
    written = np.zeros(len(slots), dtype=bool)
    cpool, cwritten = np.zeros(max(1, int(rp[-1])), dtype=dt), np.zeros(max(1, int(rp[-1])), dtype=bool)
    order = np.argsort(e["height"], kind="stable")
    leaves = {}
    for k in order:
        c0, w, nb = blk[k], blk[k + 1] - blk[k], rp[k + 1] - rp[k]
        h, ld = w + nb, e["ldw"][k]
        P = F[e["f_off"][k]:e["f_off"][k] + h * ld].reshape(h, ld)[:, :w]
        t = 1 << int(e["tgs"][k])   # LU that exchanged rows inside its tiles: block lower triangular, dense t x t diagonal tiles
        assert np.all(P[:w][(np.arange(w)[None, :] // t) > (np.arange(w)[:, None] // t)] == 0.0) if e["tgs"][k] else np.all(np.triu(P[:w], 1) == 0.0)
        nc = e["nchild"][k]
        gath = slots[e["s_off"][k]:e["s_off"][k] + nc * h].reshape(nc, h).sum(axis=0) if nc else np.zeros(h, dtype=dt)
        if compact:
            cp = e["cptr"][e["c_off"][k]:e["c_off"][k] + h + 1]
            assert cp[0] == 0 and cp[h] == sum(rp[c + 1] - rp[c] for c in range(nblk) if e["pcs_off"][c] == e["cs_off"][k] and rp[c + 1] > rp[c]) or not nc
            gath = np.array([cpool[e["cs_off"][k] + cp[i]:e["cs_off"][k] + cp[i + 1]].sum() for i in range(h)], dtype=dt) if nc else np.zeros(h, dtype=dt)
        f = b[e["perm"][c0:c0 + w]] - gath[:w]
        if use_leaves and e["lb_off"][k] >= 0:
            assert nc == 0
            L = leaves[k] = _leaf_blob(pool, e["lb_off"][k], w, ld * cs, nb, e["lb_nnzr"][k], e["lb_nnzc"][k], cs)
            assert L["srptr"][nb] == e["lb_nnzr"][k] and L["scptr"][w] == e["lb_nnzc"][k]
            z = L["W"] @ f
            y[c0:c0 + w] = z
            u = np.array([(L["srval"][L["srptr"][i]:L["srptr"][i + 1]] * z[L["srcol"][L["srptr"][i]:L["srptr"][i + 1]]]).sum() for i in range(nb)], dtype=dt)
        else:
            t = P @ f
            y[c0:c0 + w] = t[:w]
            u = t[w:] + gath[w:]
        if nb:
            where = e["ps_off"][k] + e["rel"][e["u_off"][k]:e["u_off"][k] + nb]
            assert e["ps_off"][k] >= 0 and not written[where].any() and len(set(where)) == nb   # every slot entry has ONE writer
            written[where] = True
            slots[where] = u
            cwhere = e["pcs_off"][k] + e["crel"][e["u_off"][k]:e["u_off"][k] + nb]
            assert e["pcs_off"][k] >= 0 and not cwritten[cwhere].any() and len(set(cwhere)) == nb
            cwritten[cwhere] = True
            cpool[cwhere] = u
    for k in order[::-1]:
        c0, w, nb = blk[k], blk[k + 1] - blk[k], rp[k + 1] - rp[k]
        h, ld = w + nb, e["ldw"][k]
        if k in leaves:
            L = leaves[k]
            tt = np.array([(L["scval"][L["scptr"][c]:L["scptr"][c + 1]] * x[L["scrow"][L["scptr"][c]:L["scptr"][c + 1]]]).sum() for c in range(w)], dtype=dt)
            x[c0:c0 + w] = y[c0:c0 + w] - L["W"] @ tt
            continue
        P = G[e["f_off"][k]:e["f_off"][k] + h * ld].reshape(h, ld)[:, :w]
        assert np.all(np.triu(P[:w], 1) == 0.0)
        v = np.concatenate([y[c0:c0 + w] * (dinv[c0:c0 + w] if dinv is not None else 1.0), -x[e["rows"][rp[k]:rp[k + 1]]]])
        x[c0:c0 + w] = P.T @ v
    out = np.zeros(n, dtype=dt)
    out[e["perm"]] = x
    _replay.leaves = len(leaves)
    return out


@pytest.mark.parametrize("kind", ["chol", "ldlt", "lu", "full_symmetric"])
def test_host_factorisation(kind):
    A = _poisson3d(9)
    n = A.shape[0]
    rng = np.random.default_rng(0)
    if kind == "chol":
        Ain, sym, spd, want = sp.tril(A).tocsr(), True, True, 0
    elif kind == "ldlt":
        A = (A - 1.7 * sp.identity(n)).tocsr()
        Ain, sym, spd, want = sp.tril(A).tocsr(), True, False, 1
    elif kind == "lu":
        A = (A + sp.diags(rng.random(n)) + 0.3 * sp.triu(A, 1)).tocsr()
        Ain, sym, spd, want = A, False, False, 2
    else:  # full storage, symmetric values: detected and factorised as symmetric (like the reference's 2-D example)
        Ain, sym, spd, want = A, False, True, 0
    Ain.sort_indices()
    S = hpddm.Subdomain(host_only=1)
    S.numfact(n, Ain.indptr, Ain.indices, Ain.data, sym=sym, spd=spd)
    info = S.info()
    assert info["kind"] == want and info["n"] == n
    # exact structural nnz(L) against a dense symbolic elimination
    perm = S.export("perm")
    B = (abs(A) + abs(A.T)).toarray()[np.ix_(perm, perm)] != 0
    for k in range(n):
        r = np.nonzero(B[k + 1:, k])[0] + k + 1
        B[np.ix_(r, r)] = True
    assert info["nnz_L"] == int(np.tril(B).sum())
    b = rng.random(n)
    x = _replay(S, b)
    assert np.linalg.norm(A @ x - b) / np.linalg.norm(b) < 1e-11
    assert _replay.leaves > 0   # the leaves of the nested dissection went through their blobs (W = inv(A_JJ), sparse couplings) ...
    xp = _replay(S, b, use_leaves=False)   # ... and through their dense panels: the same solution
    assert _replay.leaves == 0 and np.linalg.norm(x - xp) / np.linalg.norm(x) < 1e-12
    xc = _replay(S, b, use_leaves=False, compact=True)   # the compact hand-over of the 16-column engine: the same sums in the same order
    assert np.array_equal(xc, xp)
    S.destroy()


def _stokes2d(N):
    """MAC-like saddle point [A B^T; B 0]: two Laplacians, a discrete divergence, one pressure row dropped (constant null space)"""
    I = sp.identity(N)
    Tm = sp.diags([-1, 2, -1], [-1, 0, 1], shape=(N, N))
    L = (sp.kron(Tm, I) + sp.kron(I, Tm)).tocsr()
    D = sp.diags([-1, 1], [0, 1], shape=(N, N))
    B = sp.hstack([sp.kron(I, D), sp.kron(D, I)]).tocsr()[:-1]
    return sp.bmat([[sp.block_diag([L, L]), B.T], [B, None]]).tocsr()


def _row_swapped_pairs(n):
    """two unknowns per grid node, the two ROWS of every node exchanged: zero diagonal entries everywhere, the entries that can
    serve as pivots sit right next to the diagonal -- inside the node, hence inside the 64-column tile (supernodes of such matrices
    have even widths); unsymmetric coupling between the nodes"""
    K = _poisson3d(n)
    C = (K + 0.3 * sp.triu(K, 1)).tocsr()
    Dg = sp.diags(C.diagonal())
    M = sp.kron(Dg, np.diag([2.0, 3.0])) + sp.kron(C - Dg, np.array([[1.0, 0.5], [0.3, 1.0]]))
    A = (sp.kron(sp.identity(n ** 3), np.array([[0.0, 1.0], [1.0, 0.0]])) @ M).tocsr()
    A.eliminate_zeros()
    return A


def test_host_lu_pivots_inside_the_diagonal_tiles():
    """numeric_host.cpp / dense_host.hpp: threshold partial pivoting among the rows of a 64-column tile (static structure), the
    fall-back L D L^T -> LU on a collapsed pivot, zero diagonal entries paired with a neighbour by the ordering -- the reference's
    local solvers pivot (include/HPDDM_MUMPS.hpp:228-291); replay of the solve in numpy on the exported panels"""
    rng = np.random.default_rng(2)
    cases = [("antidiagonal", sp.csr_matrix(np.array([[0.0, 1.0], [1.0, 0.0]])), 2, True),
             ("tiny diagonal block", sp.block_diag([_poisson3d(4), sp.csr_matrix(np.array([[1e-18, 1.0], [2.0, 1e-18]]))]).tocsr(), 2, True),
             ("row-swapped pairs", _row_swapped_pairs(7), 2, True),
             ("stokes 24", _stokes2d(24), 1, False),      # paired by the ordering: L D L^T goes through without an exchange
             ("stokes 96", _stokes2d(96), None, None)]
    for name, A, kind, swapped in cases:
        A = A.tocsr()
        A.sort_indices()
        n = A.shape[0]
        S = hpddm.Subdomain(host_only=1)
        S.numfact(n, A.indptr, A.indices, A.data, sym=False)
        if kind is not None:
            assert S.info()["kind"] == kind, name
            assert bool(np.any(S.export("tgs") != 0)) == swapped, name
        b = rng.random(n)
        x = _replay(S, b)
        assert np.abs(A @ x - b).max() <= 1e-10 * max(1.0, np.abs(x).max()) * abs(A).sum(axis=1).max(), name
        S.destroy()
    # a tile without any pivot left is refused, loudly: singular matrix
    M = sp.block_diag([_poisson3d(3), sp.csr_matrix(np.array([[1.0, 1.0], [1.0, 1.0]]))]).tocsr()
    S = hpddm.Subdomain(host_only=1)
    with pytest.raises(_lib.HpddmHipError, match="pivot"):
        S.numfact(M.shape[0], M.indptr, M.indices, M.data, sym=False)
    S.destroy()
    # the plain factor (CPU baseline of bench.py) does not exist for a front whose rows were exchanged: the factorisation stands (and
    # solves, replayed below), the EXPORT of the plain panels is what is refused
    A = _row_swapped_pairs(4)
    S = hpddm.Subdomain(host_only=1, keep_plain=1)
    S.numfact(A.shape[0], A.indptr, A.indices, A.data, sym=False)
    assert np.any(S.export("tgs") != 0)
    b = rng.random(A.shape[0])
    x = _replay(S, b)
    assert np.abs(A @ x - b).max() <= 1e-10 * max(1.0, np.abs(x).max()) * abs(A).sum(axis=1).max()
    for which in ("Lplain", "Uplain"):
        with pytest.raises(_lib.HpddmHipError, match="plain factor"):
            S.export(which)
    S.destroy()
    # ... and without exchanges the plain panels of an LU factorisation are there
    K = (_poisson3d(5) + 0.2 * sp.triu(_poisson3d(5), 1)).tocsr()
    K.sort_indices()
    S = hpddm.Subdomain(host_only=1, keep_plain=1)
    S.numfact(K.shape[0], K.indptr, K.indices, K.data, sym=False)
    assert S.info()["kind"] == 2 and not np.any(S.export("tgs") != 0)
    assert S.export("Lplain").size == S.export("F").size and S.export("Uplain").size == S.export("F").size
    S.destroy()


def test_complex_host_factorisation_and_pivoting():
    """complex scalars on the host levels (numeric_host.cpp with T = std::complex<double>): complex symmetric L D L^T with plain
    transposes, general complex LU, and LU with rows exchanged inside the tiles -- numpy replay of the sweeps on the exported
    (re, im) panels against the matrix itself"""
    rng = np.random.default_rng(9)
    K = _poisson3d(8)
    n = K.shape[0]
    cases = [("complex symmetric", (K - 1.3 * sp.identity(n) + 0.4j * sp.identity(n)).tocsr(), True, 1, False),
             ("general complex", (K + 0.2 * sp.triu(K, 1) + 0.3j * sp.diags(rng.random(n))).tocsr(), False, 2, False),
             ("rows exchanged", (_row_swapped_pairs(6) * (1.0 + 0.3j)).tocsr(), False, 2, True)]
    for name, A, sym, kind, swapped in cases:
        M = sp.tril(A).tocsr() if sym else A.tocsr()
        M.sort_indices()
        m = A.shape[0]
        S = hpddm.Subdomain(host_only=1)
        S.numfact(m, M.indptr, M.indices, M.data.astype(np.complex128), sym=sym)
        assert S.info()["kind"] == kind, name
        assert bool(np.any(S.export("tgs") != 0)) == swapped, name
        b = rng.random(m) + 1j * rng.random(m)
        x = _replay(S, b)
        assert np.abs(A @ x - b).max() <= 1e-10 * max(1.0, np.abs(x).max()) * abs(A).sum(axis=1).max(), name
        S.destroy()


def test_zero_diagonals_are_paired_by_the_ordering_and_the_analysis_is_reused():
    """match_zero_diagonals (numeric_host.cpp): every unknown with a zero diagonal entry that nested dissection would eliminate
    before all of its neighbours is moved right behind a partner of its own; the rule reads the pattern and WHICH diagonal entries
    are zero only, so a refactorisation with other values reuses the analysis, and a matrix whose zeros moved is analysed again"""
    A = _stokes2d(20).tocsr()
    A.sort_indices()
    n = A.shape[0]
    S = hpddm.Subdomain(host_only=1)
    S.numfact(n, A.indptr, A.indices, A.data, sym=False)
    perm = S.export("perm")
    assert sorted(perm.tolist()) == list(range(n))                       # a permutation
    blk = S.export("blk_ptr")
    assert blk[0] == 0 and blk[-1] == n and np.all(np.diff(blk) > 0)     # contiguous, non-empty supernodes
    pos = np.empty(n, dtype=np.int64)
    pos[perm] = np.arange(n)
    zero = np.flatnonzero(A.diagonal() == 0.0)
    assert len(zero) > 100
    Acsr = A.tocsr()
    later = 0
    for i in zero:   # a pressure is never the first of its neighbourhood
        nb = Acsr.indices[Acsr.indptr[i]:Acsr.indptr[i + 1]]
        nb = nb[nb != i]
        later += int(pos[i] > pos[nb].min())
    assert later == len(zero)
    b = np.random.default_rng(5).random(n)
    assert np.abs(A @ _replay(S, b) - b).max() < 1e-10
    t_order = S.info()["t_order"]
    B = (A * 3.0).tocsr()                                                # same pattern, same zeros: numerical phase only
    S.numfact(n, B.indptr, B.indices, B.data, sym=False)
    assert S.info()["t_order"] == t_order and np.array_equal(S.export("perm"), perm)
    assert np.abs(B @ _replay(S, b) - b).max() < 1e-10
    # zero diagonal entries that are STORED: the same rule; and when they stop being zero on the same pattern, another analysis
    co = A.tocoo()
    E = sp.csr_matrix((np.concatenate([co.data, np.zeros(len(zero))]), (np.concatenate([co.row, zero]), np.concatenate([co.col, zero]))), shape=A.shape)
    E.sort_indices()
    assert E.nnz == A.nnz + len(zero)
    S.numfact(n, E.indptr, E.indices, E.data, sym=False)
    perm_e = S.export("perm")
    pos[perm_e] = np.arange(n)
    assert all(pos[i] > pos[np.setdiff1d(E.indices[E.indptr[i]:E.indptr[i + 1]], [i])].min() for i in zero)
    assert np.abs(A @ _replay(S, b) - b).max() < 1e-10
    C = E.copy()
    C.data[C.data == 0.0] = -0.05
    assert np.array_equal(C.indices, E.indices) and np.count_nonzero(C.diagonal() == 0.0) == 0
    S.numfact(n, C.indptr, C.indices, C.data, sym=False)
    assert not np.array_equal(S.export("perm"), perm_e)                   # same pattern, other zeros: not the same analysis
    assert np.abs(C @ _replay(S, b) - b).max() < 1e-10
    S.destroy()


def test_ordering_handles_disconnected_and_tiny_graphs():
    for M in (sp.identity(5).tocsr(), sp.block_diag([_poisson3d(3), _poisson3d(2), sp.identity(3)]).tocsr(), sp.csr_matrix(np.array([[2.0]]))):
        n = M.shape[0]
        L = sp.tril(M).tocsr()
        L.sort_indices()
        S = hpddm.Subdomain(host_only=1)
        S.numfact(n, L.indptr, L.indices, L.data, sym=True, spd=True)
        b = np.arange(1.0, n + 1)
        assert np.allclose(M @ _replay(S, b), b)
        S.destroy()


def test_no_cpu_fallback():
    """without a GPU every compute entry point must fail loudly"""
    if hpddm.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_lib.HpddmHipError):
        hpddm.require_device()
    A = _poisson3d(4)
    L = sp.tril(A).tocsr()
    S = hpddm.Subdomain()
    with pytest.raises(_lib.HpddmHipError):
        S.numfact(A.shape[0], L.indptr, L.indices, L.data, sym=True, spd=True)  # upload needs the device
    S2 = hpddm.Subdomain(host_only=1)
    S2.numfact(A.shape[0], L.indptr, L.indices, L.data, sym=True, spd=True)
    with pytest.raises(_lib.HpddmHipError):
        S2.solve(np.ones(A.shape[0]))


def test_c_api_shim_exports_the_reference_names():
    """libhpddm_c_hip.so (built by oracle/Makefile.ref where MPI is available) exports every function include/hpddm_c_compat.h
    declares -- the names of the reference's interface/HPDDM.h"""
    so = os.path.join(ROOT, "hpddm_amd", "libhpddm_c_hip.so")
    if not os.path.exists(so):
        pytest.skip("libhpddm_c_hip.so not built (needs MPI: make -C oracle ref)")
    header = open(os.path.join(ROOT, "include", "hpddm_c_compat.h")).read()
    declared = set(re.findall(r"\b(Hpddm\w+|nrm2|axpy)\s*\(", header))
    assert len(declared) >= 30
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    missing = sorted(declared - exported)
    assert not missing, missing
    # ... and so does the complex build of the same source (K = double _Complex: one scalar type per library, HPDDM.h:34-50)
    soz = os.path.join(ROOT, "hpddm_amd", "libhpddm_c_hip_z.so")
    if os.path.exists(soz):
        out = subprocess.run(["nm", "-D", "--defined-only", soz], capture_output=True, text=True).stdout
        assert not sorted(declared - {ln.split()[-1] for ln in out.splitlines() if ln.strip()})


def test_plain_c_client_of_the_header(tmp_path):
    """include/hpddm_hip.h is a C header: examples/c_abi_host.c (C99, -Wall -Werror -pedantic) compiles against it, links with the
    library and runs the host-only part of the life cycle on CPU, matrix dumps included"""
    import subprocess
    from hpddm_amd.matrix_io import read_matrix
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "c_abi_host")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "c_abi_host.c"),
                           "-o", exe, "-L" + os.path.join(root, "hpddm_amd"), "-lhpddm_hip", "-lm", "-Wl,-rpath," + os.path.join(root, "hpddm_amd")])
    res = subprocess.run([exe, str(tmp_path / "dump")], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0 and res.stdout.startswith("ok"), res.stdout + res.stderr
    for s in range(2):
        mat = read_matrix(str(tmp_path / f"dump_{s}_2.txt"))
        assert (mat["n"], mat["nnz"], mat["sym"]) == (6, 16, False) and np.allclose(mat["a"][mat["ja"] == np.repeat(np.arange(6), np.diff(mat["ia"]))], 2.0)


@pytest.mark.parametrize("problem,limit", [("poisson24", 1.46e6), ("elasticity10", 5.9e5)])
def test_fill_of_the_ordering_does_not_regress(problem, limit):
    """nnz(L) is the algorithmic byte count of every SpTRSV: guard the ordering (nested dissection with supervariable
    compression, multilevel bisection for the denser graphs) against regressions, and against a minimum-degree baseline (SuperLU's
    MMD on A + A^T, the better of its two orderings on these matrices)."""
    import scipy.sparse.linalg as spl
    from hpddm_amd.generate import generate3d, generate_elasticity3d
    from oracle.ras_oracle import csr_full
    sd = generate3d(24, 1, 0, sym=True)[0] if problem == "poisson24" else generate_elasticity3d(10, 1, 0)[0]
    S = hpddm.Subdomain(host_only=1)
    S.numfact(sd["n"], sd["ia"], sd["ja"], sd["a"], sym=sd["sym"], spd=True)
    info = S.info()
    assert info["nnz_L"] <= limit, info
    assert info["stored"] <= 1.18 * info["nnz_L"]            # padding of the panel layout
    mmd = spl.splu(csr_full(sd).tocsc(), permc_spec="MMD_AT_PLUS_A").L.nnz
    assert info["nnz_L"] <= 0.75 * mmd
    S.destroy()


@pytest.mark.parametrize("seed,n,density", [(0, 150, 0.02), (1, 220, 0.008), (2, 90, 0.06), (3, 300, 0.004)])
def test_exact_fill_count_on_unstructured_graphs(seed, n, density):
    """nnz(L) of the analysis (column counts by skeleton matrix + least common ancestors, symbolic.cpp) against a dense symbolic
    elimination on random sparse SPD matrices: several components, isolated vertices, irregular degrees"""
    rng = np.random.default_rng(seed)
    T = sp.triu(sp.random(n, n, density=density, random_state=seed, format="csr"), 1).tocsr()
    T.data[:] = -rng.random(T.nnz)
    A = (T + T.T).tocsr()
    A = (A + sp.diags(np.asarray(abs(A).sum(axis=1)).ravel() + 1.0)).tocsr()   # strictly diagonally dominant: SPD
    L = sp.tril(A).tocsr()
    L.sort_indices()
    S = hpddm.Subdomain(host_only=1)
    S.numfact(n, L.indptr, L.indices, L.data, sym=True, spd=True)
    perm = S.export("perm")
    B = (abs(A) + abs(A.T)).toarray()[np.ix_(perm, perm)] != 0
    for k in range(n):
        r = np.nonzero(B[k + 1:, k])[0] + k + 1
        B[np.ix_(r, r)] = True
    assert S.info()["nnz_L"] == int(np.tril(B).sum())
    b = rng.random(n)
    assert np.allclose(A @ _replay(S, b), b)
    S.destroy()
