"""Drop-in check: the UNCHANGED reference programs, compiled around our local solver through the Solver<K> plug-in
(include/hpddm_hip_sub.hpp, -DSUBDOMAIN=HPDDM::HipSub; recipe oracle/Makefile.ref), run under MPI on the GPU box and
reproduce the reference's own results (iteration counts of BASELINE.md section 2)."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
MPIEXEC = "/opt/conda/bin/mpiexec"


def _have():
    return os.path.exists(os.path.join(REF, "schwarz_hipsub")) and os.path.exists(MPIEXEC)


def _run(np_, args, exe="schwarz_hipsub"):
    env = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL", HPDDM_HIP_NUM_THREADS="4")
    cmd = [MPIEXEC, "-n", str(np_), os.path.join(REF, exe)] + args.split()
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    assert "BUG HipSub" not in res.stderr, res.stderr[-2000:]
    return res.stdout


@pytest.mark.skipif(not _have(), reason="oracle/_ref/schwarz_hipsub not built (needs /root/reference at build time)")
@pytest.mark.parametrize("args,its,resid", [
    ("-hpddm_verbosity=1 -Nx 40 -Ny 40 -hpddm_gmres_restart=25 -hpddm_max_it 80", 19, 1.428088e-05),
    ("-hpddm_verbosity=1 -Nx 40 -Ny 40 -symmetric_csr=1 -hpddm_operator_spd", 19, 1.747862e-05),
    ("-hpddm_verbosity=1 -Nx 40 -Ny 40 -hpddm_schwarz_coarse_correction deflated -hpddm_geneo_nu=0", 17, 3.415834e-05),
    ("-hpddm_verbosity=1 -Nx 200 -Ny 200", 45, 1.660748e-04),
])
def test_unchanged_schwarz_cpp_with_hip_local_solver(args, its, resid):
    out = _run(4, args)
    m = re.search(r"GMRES converges after (\d+) iteration", out)
    assert m and int(m.group(1)) == its, out[-1500:]
    r = re.search(r"--- residual = (\S+) / (\S+)", out)
    assert r and abs(float(r.group(1)) - resid) <= 5e-4 * resid, out[-500:]


@pytest.mark.skipif(not _have(), reason="oracle/_ref not built")
def test_unchanged_single_rank_direct_solve_and_local_solver_benchmark(tmp_path):
    mat = str(tmp_path / "mat.txt")
    out = _run(1, f"-Nx 40 -Ny 40 -hpddm_dump_matrices={mat}")
    r = re.search(r"--- residual = (\S+) / (\S+)", out)
    assert r and float(r.group(1)) / float(r.group(2)) <= 1e-6  # examples/schwarz.cpp:178
    assert os.path.exists(mat)
    # benchmark/local_solver.cpp:92-127 protocol: one line per trial, seconds for nu = 1, 2, 4
    out = _run(1, f"{mat} -rhs=4 -solve_phase_only=1", exe="local_solver_hipsub")
    rows = [re.findall(r"\d\.\d{5}e[-+]\d{2}", ln) for ln in out.strip().splitlines() if re.match(r"^\s*\d\.\d+e[-+]\d+", ln)]
    assert len(rows) == 3 and all(len(r) == 3 for r in rows), out  # 3 trials x (nu = 1, 2, 4)


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "schwarz_hipsub_z")), reason="oracle/_ref/schwarz_hipsub_z not built")
@pytest.mark.parametrize("args,pat,its,resid", [
    ("-hpddm_verbosity=1 -Nx 40 -Ny 40", "GMRES", 19, 1.428088e-05),
    ("-hpddm_verbosity=1 -Nx 40 -Ny 40 -symmetric_csr=1", "GMRES", 19, None),
])
def test_unchanged_schwarz_cpp_complex_scalars(args, pat, its, resid):
    """K = std::complex<double> (examples/schwarz.hpp -DFORCE_COMPLEX): HipSub<std::complex<double>> behind the unchanged
    driver; the complex LAPACK build of the reference (oracle/_ref/schwarz_cpp_z) gives 19 iterations, 1.428088e-05"""
    out = _run(4, args, exe="schwarz_hipsub_z")
    m = re.search(pat + r" converges after (\d+) iteration", out)
    assert m and int(m.group(1)) == its, out[-1500:]
    r = re.search(r"--- residual = (\S+) / (\S+)", out)
    assert r and float(r.group(1)) / float(r.group(2)) <= 1e-6
    if resid:
        assert abs(float(r.group(1)) - resid) <= 5e-4 * resid, out[-500:]


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "schwarz_c_hip")), reason="oracle/_ref/schwarz_c_hip not built")
@pytest.mark.parametrize("args,its,resid", [
    ("-hpddm_verbosity=1 -Nx 40 -Ny 40", 19, 1.428088e-05),
    ("-hpddm_verbosity=1 -Nx 40 -Ny 40 -symmetric_csr=1 -hpddm_operator_spd", 19, 1.747862e-05),
    ("-hpddm_verbosity=1 -Nx 40 -Ny 40 -hpddm_schwarz_coarse_correction deflated -hpddm_geneo_nu=0", 17, 3.415834e-05),
    ("-hpddm_verbosity=1 -Nx 40 -Ny 40 -generate_random_rhs 3 -hpddm_krylov_method bgmres", None, None),
    ("-hpddm_verbosity=1 -Nx 200 -Ny 200", 45, 1.660748e-04),
])
def test_unchanged_c_example_against_the_c_api_shim(args, its, resid):
    """examples/schwarz.c + generate.c (written against interface/HPDDM.h) linked with libhpddm_c_hip.so instead of
    interface/hpddm_c.cpp: 4 MPI ranks, one subdomain each, halo and reductions through MPI, every solve on the GPU.
    The C++ example of the reference gives the same iteration counts / residuals."""
    out = _run(4, args, exe="schwarz_c_hip")
    r = re.findall(r"residual = (\S+) / (\S+)|^\s+(\S+) / (\S+) \(rhs", out, re.M)
    assert r, out[-1500:]
    if its is not None:
        m = re.search(r"GMRES converges after (\d+) iteration", out)
        assert m and int(m.group(1)) == its, out[-1500:]
        first = [v for v in r[0] if v]
        assert abs(float(first[0]) - resid) <= 5e-4 * resid, out[-500:]
    else:
        assert re.search(r"BGMRES converges after \d+ iteration", out), out[-1500:]
        for grp in r:
            vals = [float(v) for v in grp if v]
            assert vals[0] / vals[1] <= 1e-2   # the example's own check (examples/schwarz.c:120)


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "schwarz_c_hip_z")), reason="oracle/_ref/schwarz_c_hip_z not built")
@pytest.mark.parametrize("args,pat,its,resid", [
    ("-hpddm_verbosity=1 -Nx 40 -Ny 40", "GMRES", 19, 1.428088e-05),
    ("-hpddm_verbosity=1 -Nx 40 -Ny 40 -symmetric_csr=1", "GMRES", 19, None),
    ("-hpddm_verbosity=1 -Nx 40 -Ny 40 -hpddm_schwarz_coarse_correction deflated -hpddm_geneo_nu=0", "GMRES", 17, 3.208441e-05),
    ("-hpddm_verbosity=1 -Nx 40 -Ny 40 -generate_random_rhs 3 -hpddm_krylov_method bgmres", "BGMRES", None, None),
])
def test_unchanged_c_example_complex_scalars_against_the_c_api_shim(args, pat, its, resid):
    """interface/HPDDM.h:34-50: the C library is built for ONE scalar type; libhpddm_c_hip_z.so is the shim compiled with
    -DFORCE_COMPLEX (K = double _Complex), examples/schwarz.c + generate.c compiled the same way and linked with it, unchanged.
    Expected counts and residuals: oracle/_ref/schwarz_cpp_z, the reference's own complex build (dense LAPACK local solver)."""
    out = _run(4, args, exe="schwarz_c_hip_z")
    m = re.search(pat + r" converges after (\d+) iteration", out)
    assert m, out[-1500:]
    r = re.findall(r"residual = (\S+) / (\S+)|^\s+(\S+) / (\S+) \(rhs", out, re.M)
    assert r, out[-1500:]
    if its is not None:
        assert int(m.group(1)) == its, out[-1500:]
    if resid:
        first = [v for v in r[0] if v]
        assert abs(float(first[0]) - resid) <= 5e-4 * resid, out[-500:]
    for grp in r:
        vals = [float(v) for v in grp if v]
        assert vals[0] / vals[1] <= 1e-2   # the example's own check (examples/schwarz.c:120)


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "schwarz_c_hip")), reason="oracle/_ref/schwarz_c_hip not built")
def test_unchanged_c_example_single_rank_direct_solve():
    out = _run(1, "-Nx 40 -Ny 40", exe="schwarz_c_hip")
    r = re.search(r"--- residual = (\S+) / (\S+)", out)
    assert r and float(r.group(1)) / float(r.group(2)) <= 1e-6


def _parse_dump(path):
    import numpy as np
    out, lines, i = {}, open(path).read().split("\n"), 0
    while i < len(lines):
        if lines[i].startswith("@"):
            name, kind, cnt = lines[i][1:].split()
            cnt = int(cnt)
            if kind == "z":   # complex K: "re im" pairs
                out[name] = np.array([complex(*map(float, ln.split())) for ln in lines[i + 1:i + 1 + cnt]])
            else:
                out[name] = np.array(lines[i + 1:i + 1 + cnt], dtype=np.float64 if kind == "f" else np.int32)
            i += 1 + cnt
        else:
            i += 1
    return out


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "ref_harness_hipcc")), reason="oracle/_ref/ref_harness_hipcc not built")
@pytest.mark.parametrize("name,ranks,mu,its", [("p40_deflated", 4, 1, 17), ("p30_6ranks_deflated_nu3", 6, 2, None), ("p40_bgmres_deflated_mu2", 4, 2, None)])
def test_coarse_correction_hook_runs_the_deflation_on_the_device(name, ranks, mu, its, tmp_path):
    """Boundary B2: Preconditioner::cc_ = HPDDM::HipCoarseCorrection (include/hpddm_hip_coarse.hpp) inside the reference's own
    two-level Schwarz object -- local solves through HipSub, the two contractions of Schwarz::deflation through the panel
    kernels, coarse solve and halo by the reference.  Same deflation output, preconditioner apply, iteration count and solution
    as the pure reference run the golden fixture was dumped from (17 iterations on the 40 x 40 case)."""
    import numpy as np
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    _run(ranks, f"-out {tmp_path} -case hook -mu {mu} -hpddm_verbosity=1 " + str(g["options"]), exe="ref_harness_hipcc")
    for r in range(ranks):
        d = _parse_dump(os.path.join(tmp_path, f"hook_r{r}.txt"))
        for key, tol in (("deflation_out", 1e-11), ("apply_out", 1e-9), ("sol", 1e-6)):
            ref = g[f"{key}_r{r}"]
            assert np.abs(d[key] - ref).max() <= tol * max(1e-300, np.abs(ref).max()), (key, r)
        assert int(d["iterations"][0]) == int(g["iterations_r0"][0]) and (its is None or int(d["iterations"][0]) == its)


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "ref_harness_hipcc_z")), reason="oracle/_ref/ref_harness_hipcc_z not built")
@pytest.mark.parametrize("name,ranks,mu", [("z_p30_6ranks_deflated_nu3", 6, 2), ("z_p30_gmres_left_deflated", 4, 1)])
def test_coarse_correction_hook_complex(name, ranks, mu, tmp_path):
    """Boundary B2 with K = std::complex<double> (-DFORCE_COMPLEX build of the reference): HipCoarseCorrection on
    HpddmHipPanelCreateZ -- uc = Z^H (D in) and Z y on the device with the complex vectors as they are, local solves through
    HipSub<std::complex<double>> -- against the fixtures dumped from the pure reference"""
    import numpy as np
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    _run(ranks, f"-out {tmp_path} -case hookz -mu {mu} -hpddm_verbosity=1 " + str(g["options"]), exe="ref_harness_hipcc_z")
    for r in range(ranks):
        d = _parse_dump(os.path.join(tmp_path, f"hookz_r{r}.txt"))
        for key, tol in (("deflation_out", 1e-11), ("apply_out", 1e-9), ("sol", 1e-6)):
            ref = g[f"{key}_r{r}"]
            got = d[key]
            assert np.abs(got - ref).max() <= tol * max(1e-300, np.abs(ref).max()), (key, r)
        assert int(d["iterations"][0]) == int(g["iterations_r0"][0])


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "custom_operator_c_hip")), reason="oracle/_ref/custom_operator_c_hip not built")
@pytest.mark.parametrize("ranks,args,method,its", [
    (1, "", "GMRES", 6),
    (2, "-n 300 -mu 1", "GMRES", 6),
    (3, "-n 57 -mu 4 -hpddm_krylov_method bgmres", "BGMRES", 5),
    (2, "-n 200 -mu 1 -hpddm_krylov_method cg", "CG", 7),
    (2, "-n 80 -mu 3 -hpddm_krylov_method bcg", "BCG", 6),
    (4, "-n 50 -mu 2 -hpddm_variant left", "GMRES", 7),
    (2, "-n 120 -mu 1 -hpddm_krylov_method gcrodr -hpddm_recycle 5 -hpddm_gmres_restart 10 -hpddm_tol 1e-10", "GCRODR", 10),
])
def test_unchanged_custom_operator_example_against_the_c_api_shim(ranks, args, method, its):
    """examples/custom_operator.c (HpddmCustomOperatorSolve, interface/HPDDM.h:115: operator and preconditioner as callbacks, a
    tridiagonal matrix with its Jacobi preconditioner) linked with libhpddm_c_hip.so instead of interface/hpddm_c.cpp.  The
    expected iteration counts are those of the same source linked with the reference's own interface/hpddm_c.cpp
    (oracle/_ref/custom_operator_c_ref, run in the build container); the right-hand sides of the example are seeded by the clock,
    which moves the count by at most one in both builds."""
    out = _run(ranks, args, exe="custom_operator_c_hip")
    m = re.search(method + r" converges after (\d+) iteration", out)
    assert m and abs(int(m.group(1)) - its) <= 1, out[-1500:]
