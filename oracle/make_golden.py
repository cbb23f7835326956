#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the COMPILED REFERENCE (oracle/_ref/ref_harness, built by Makefile.ref).

Test infrastructure only.  Runs in the build container (needs /root/reference for the build, MPICH + MKL from
/opt/conda for the run); the resulting fixtures are DATA (inputs + the reference's outputs) and are committed so the
GPU box, which has no /root/reference, can check parity against them.

Each fixture holds, per MPI rank r (= subdomain r): CSR matrix (ia_r, ja_r, a_r), generator weights d_in_r, the
partition of unity d_r after Schwarz::multiplicityScaling, the neighbour map (neighbors_r, map_r_k), the RHS block
f_r, and the reference's outputs of exchange / GMV / Solver::solve / apply / deflation, the GMRES iteration count,
the solution and computeResidual's two numbers; plus the residual history parsed from the reference's own log.
"""
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "_ref")
OUT = os.path.join(HERE, "..", "tests", "golden")

# name, ranks, harness mu, reference options
CASES = [
    ("p40_onelevel", 4, 1, "-Nx 40 -Ny 40 -hpddm_gmres_restart=25 -hpddm_max_it 80"),
    ("p40_onelevel_mu3", 4, 3, "-Nx 40 -Ny 40"),
    ("p40_sym_spd", 4, 1, "-Nx 40 -Ny 40 -symmetric_csr=1 -hpddm_operator_spd"),
    ("p40_sym_ldlt", 4, 1, "-Nx 40 -Ny 40 -symmetric_csr=1"),
    ("p40_overlap2", 4, 2, "-Nx 40 -Ny 40 -overlap 2"),
    ("p40_deflated", 4, 1, "-Nx 40 -Ny 40 -hpddm_schwarz_coarse_correction deflated -hpddm_geneo_nu=0"),
    ("p40_deflated_mu2_ov2", 4, 2, "-Nx 40 -Ny 40 -overlap 2 -hpddm_schwarz_coarse_correction deflated -hpddm_geneo_nu=0"),
    ("p40_additive", 4, 1, "-Nx 40 -Ny 40 -hpddm_schwarz_coarse_correction additive -hpddm_geneo_nu=0"),
    ("p40_balanced", 4, 1, "-Nx 40 -Ny 40 -hpddm_schwarz_coarse_correction balanced -hpddm_geneo_nu=0"),
    ("p36x60_9ranks", 9, 2, "-Nx 36 -Ny 60 -overlap 2"),
    ("p30_6ranks_mgs_left", 6, 1, "-Nx 30 -Ny 30 -hpddm_orthogonalization mgs -hpddm_variant left"),
    ("p40_bgmres_mu4", 4, 4, "-Nx 40 -Ny 40 -hpddm_krylov_method bgmres"),
    ("p40_bgmres_deflated_mu2", 4, 2, "-Nx 40 -Ny 40 -hpddm_krylov_method bgmres -hpddm_schwarz_coarse_correction deflated -hpddm_geneo_nu=0 -hpddm_gmres_restart=5"),
    ("p30_6ranks_bgmres_left_mu3", 6, 3, "-Nx 30 -Ny 30 -hpddm_krylov_method bgmres -hpddm_variant left"),
    ("p40_fgmres_restart8_mu2", 4, 2, "-Nx 40 -Ny 40 -hpddm_variant flexible -hpddm_gmres_restart=8"),
    ("p40_fbgmres_mu3", 4, 3, "-Nx 40 -Ny 40 -hpddm_krylov_method bgmres -hpddm_variant flexible -hpddm_gmres_restart=6"),
    ("p40_oras_og", 4, 2, "-Nx 40 -Ny 40 -overlap 2 -hpddm_schwarz_method oras -optimized_shift 30"),
    ("p40_soras_os_sym", 4, 1, "-Nx 40 -Ny 40 -overlap 2 -symmetric_csr=1 -hpddm_schwarz_method soras -optimized_shift 30"),
    ("p40_soras_os_deflated", 4, 2, "-Nx 40 -Ny 40 -overlap 2 -hpddm_schwarz_method soras -optimized_shift 20 -hpddm_schwarz_coarse_correction deflated -hpddm_geneo_nu=0"),
    ("p40_bcg_asm_mu3", 4, 3, "-Nx 40 -Ny 40 -hpddm_krylov_method bcg -hpddm_schwarz_method asm"),
    ("p30_6ranks_bcg_asm_sym_mu2", 6, 2, "-Nx 30 -Ny 30 -symmetric_csr=1 -hpddm_krylov_method bcg -hpddm_schwarz_method asm -hpddm_operator_spd"),
    ("p40_penalized_mu2", 4, 2, "-Nx 40 -Ny 40 -penalize 1"),
    ("p40_penalized_left_mu2_ov2", 4, 2, "-Nx 40 -Ny 40 -overlap 2 -penalize 1 -hpddm_variant left"),
    ("p40_penalized_sym_left", 4, 1, "-Nx 40 -Ny 40 -symmetric_csr=1 -penalize 1 -hpddm_variant left"),
    ("p40_bgmres_rhs_deflation_mu4", 4, 4, "-Nx 40 -Ny 40 -dependent_rhs 1 -hpddm_krylov_method bgmres -hpddm_deflation_tol 1e-6"),
    ("p40_bgmres_rhs_deflation_restart_mu4", 4, 4, "-Nx 40 -Ny 40 -dependent_rhs 1 -hpddm_krylov_method bgmres -hpddm_deflation_tol 1e-4 -hpddm_gmres_restart=6"),
    # K = std::complex<double> (ref_harness_z, symCoarse 'G'): diagonal times (1 + re/100 + i im/100), complex right-hand sides
    ("z_p30_gmres_mu2", 4, 2, "-Nx 30 -Ny 30 -complex_shift_re -5 -complex_shift_im 3"),
    ("z_p30_gmres_left_deflated", 4, 1, "-Nx 30 -Ny 30 -complex_shift_re -10 -complex_shift_im 2 -hpddm_variant left -hpddm_schwarz_coarse_correction deflated -hpddm_geneo_nu=0"),
    ("z_p30_6ranks_bgmres_mu3_balanced", 6, 3, "-Nx 30 -Ny 30 -overlap 2 -complex_shift_re -5 -complex_shift_im 3 -hpddm_krylov_method bgmres -hpddm_schwarz_coarse_correction balanced -hpddm_geneo_nu=0"),
    # round 5: the additive correction, the flexible variant with restarts and flexible Block GMRES for K = std::complex<double>
    ("z_p30_additive_mu2", 4, 2, "-Nx 30 -Ny 30 -complex_shift_re -10 -complex_shift_im 2 -hpddm_schwarz_coarse_correction additive -hpddm_geneo_nu=0"),
    ("z_p30_fgmres_restart8_mu2", 4, 2, "-Nx 30 -Ny 30 -complex_shift_re -10 -complex_shift_im 2 -hpddm_variant flexible -hpddm_gmres_restart=8"),
    ("z_p30_fbgmres_mu3", 4, 3, "-Nx 30 -Ny 30 -complex_shift_re -10 -complex_shift_im 2 -hpddm_krylov_method bgmres -hpddm_variant flexible -hpddm_gmres_restart=6"),
    ("z_p30_richardson_mu2", 4, 2, "-Nx 30 -Ny 30 -complex_shift_re -10 -complex_shift_im 2 -hpddm_krylov_method richardson -hpddm_max_it 15 -hpddm_richardson_damping_factor 0.7"),
    ("z_p30_none_deflated_mu2", 4, 2, "-Nx 30 -Ny 30 -complex_shift_re -10 -complex_shift_im 2 -hpddm_krylov_method none -hpddm_schwarz_coarse_correction deflated -hpddm_geneo_nu=0"),
    ("z_p36x60_9ranks_mu2", 9, 2, "-Nx 36 -Ny 60 -overlap 2 -complex_shift_re -10 -complex_shift_im 2"),
    ("p40_osm_og", 4, 2, "-Nx 40 -Ny 40 -overlap 2 -hpddm_schwarz_method osm -optimized_shift 30"),
    ("z_p30_bgmres_mu8", 4, 8, "-Nx 30 -Ny 30 -complex_shift_re -10 -complex_shift_im 2 -hpddm_krylov_method bgmres -hpddm_gmres_restart=10"),
    ("z_p30_6ranks_deflated_nu3", 6, 2, "-Nx 30 -Ny 30 -overlap 2 -deflation_nu 3 -complex_shift_re -10 -complex_shift_im 2 -hpddm_schwarz_coarse_correction deflated -hpddm_geneo_nu=0"),
    # complex optimised local matrices (callNumfact(A_opt) with K = std::complex<double>: what ORAS does for Helmholtz), OG and OS
    ("z_p30_oras_og_mu2", 4, 2, "-Nx 30 -Ny 30 -overlap 2 -complex_shift_re -10 -complex_shift_im 2 -hpddm_schwarz_method oras -optimized_shift 30 -optimized_shift_im 20"),
    ("z_p30_soras_os_deflated", 4, 2, "-Nx 30 -Ny 30 -overlap 2 -complex_shift_re -10 -complex_shift_im 2 -hpddm_schwarz_method soras -optimized_shift 20 -optimized_shift_im 10 -hpddm_schwarz_coarse_correction deflated -hpddm_geneo_nu=0"),
    # complex Block GMRES with right-hand-side deflation (RRQR of the residual block, include/HPDDM_iterative.hpp:583-595)
    ("z_p30_bgmres_rhs_deflation_mu4", 4, 4, "-Nx 30 -Ny 30 -dependent_rhs 1 -complex_shift_re -10 -complex_shift_im 2 -hpddm_krylov_method bgmres -hpddm_deflation_tol 1e-6"),
    # several deflation vectors per subdomain (the constant one + smooth local ones, dumped as ev): coarse blocks larger than 1 x 1
    ("p30_6ranks_deflated_nu3", 6, 2, "-Nx 30 -Ny 30 -overlap 2 -deflation_nu 3 -hpddm_schwarz_coarse_correction deflated -hpddm_geneo_nu=0"),
    ("p40_bfbcg_asm_mu3", 4, 3, "-Nx 40 -Ny 40 -hpddm_krylov_method bfbcg -hpddm_schwarz_method asm -hpddm_tol 1e-4"),
    ("p40_bfbcg_asm_rhs_deflation_mu4", 4, 4, "-Nx 40 -Ny 40 -dependent_rhs 1 -hpddm_krylov_method bfbcg -hpddm_schwarz_method asm -hpddm_deflation_tol 1e-6 -hpddm_tol 1e-4"),
    # GCRO-DR: GMRES(10) recycling 4 harmonic Ritz vectors, two successive solves (the second one starts from the recycled space)
    # GCRO-DR for K = std::complex<double>: two solves (the second starts from the recycled space), right and left preconditioning
    ("z_p30_gcrodr_two_solves", 4, 1, "-Nx 30 -Ny 30 -complex_shift_re -10 -complex_shift_im 2 -second_solve 1 -hpddm_krylov_method gcrodr -hpddm_recycle 4 -hpddm_gmres_restart 10"),
    ("z_p30_gcrodr_mu2", 4, 2, "-Nx 30 -Ny 30 -complex_shift_re -10 -complex_shift_im 2 -second_solve 1 -hpddm_krylov_method gcrodr -hpddm_recycle 3 -hpddm_gmres_restart 8"),
    # (with -hpddm_schwarz_coarse_correction deflated the reference's complex GCRO-DR does not converge at all -- 100 iterations, residual
    # 2.6e+01 of 3.0e+01 on the 6-rank case its GMRES solves in 12: no fixture)
    ("z_p30_gcrodr_left_mgs", 4, 1, "-Nx 30 -Ny 30 -complex_shift_re -5 -complex_shift_im 3 -second_solve 1 -hpddm_krylov_method gcrodr -hpddm_recycle 4 -hpddm_gmres_restart 10 -hpddm_variant left -hpddm_orthogonalization mgs"),
    ("z_p30_gcrodr_target_lm_same_system", 4, 1, "-Nx 30 -Ny 30 -complex_shift_re -5 -complex_shift_im 3 -second_solve 1 -hpddm_krylov_method gcrodr -hpddm_recycle 3 -hpddm_gmres_restart 8 -hpddm_recycle_target LM -hpddm_recycle_same_system 1"),
    ("z_p30_bgcrodr_two_solves_mu2", 4, 2, "-Nx 30 -Ny 30 -complex_shift_re -10 -complex_shift_im 2 -second_solve 1 -hpddm_krylov_method bgcrodr -hpddm_recycle 3 -hpddm_gmres_restart 8"),
    ("z_p30_6ranks_bgcrodr_left_mu3", 6, 3, "-Nx 30 -Ny 30 -overlap 2 -complex_shift_re -5 -complex_shift_im 3 -second_solve 1 -hpddm_krylov_method bgcrodr -hpddm_recycle 2 -hpddm_gmres_restart 8 -hpddm_variant left"),
    # block CG methods for K = std::complex<double> on a Hermitian positive definite operator (diagonal times 1.05, complex right-hand
    # sides): the reference's CG / BCG / BFBCG converge in 17 / 20 / 15 iterations (round 4 claimed they diverge: the operator tried
    # then, -complex_shift_re -5, is indefinite)
    ("z_p30_cg_asm_hpd_mu3", 4, 3, "-Nx 30 -Ny 30 -symmetric_csr=1 -hpddm_operator_spd -complex_shift_re 5 -complex_shift_im 0 -hpddm_schwarz_method asm -hpddm_krylov_method cg"),
    ("z_p30_bcg_asm_hpd_mu3", 4, 3, "-Nx 30 -Ny 30 -symmetric_csr=1 -hpddm_operator_spd -complex_shift_re 5 -complex_shift_im 0 -hpddm_schwarz_method asm -hpddm_krylov_method bcg"),
    ("z_p30_bfbcg_asm_hpd_mu3", 4, 3, "-Nx 30 -Ny 30 -symmetric_csr=1 -hpddm_operator_spd -complex_shift_re 5 -complex_shift_im 0 -hpddm_schwarz_method asm -hpddm_krylov_method bfbcg"),
    ("z_p30_bfbcg_asm_rhs_deflation_mu4", 4, 4, "-Nx 30 -Ny 30 -symmetric_csr=1 -hpddm_operator_spd -complex_shift_re 5 -complex_shift_im 0 -dependent_rhs 1 -hpddm_schwarz_method asm -hpddm_krylov_method bfbcg -hpddm_deflation_tol 1e-6"),
    # the same operator without the shift (the plain 5-point Laplacian, complex right-hand sides): the reference's BFBCG takes 34
    # iterations.  (Its BCG meets a block it takes for rank-deficient and hands over to CG at an iteration that depends on rounding:
    # 58 iterations into the run here; not a fixture.)
    ("z_p30_bfbcg_asm_shift0_mu3", 4, 3, "-Nx 30 -Ny 30 -symmetric_csr=1 -hpddm_operator_spd -complex_shift_re 0 -complex_shift_im 0 -hpddm_schwarz_method asm -hpddm_krylov_method bfbcg"),
    ("z_p30_6ranks_bcg_asm_hpd_mu2", 6, 2, "-Nx 30 -Ny 30 -symmetric_csr=1 -hpddm_operator_spd -complex_shift_re 5 -complex_shift_im 0 -hpddm_schwarz_method asm -hpddm_krylov_method bcg"),
    ("p40_gcrodr_two_solves", 4, 1, "-Nx 40 -Ny 40 -second_solve 1 -hpddm_krylov_method gcrodr -hpddm_recycle 4 -hpddm_gmres_restart 10"),
    ("p40_gcrodr_same_system", 4, 1, "-Nx 40 -Ny 40 -second_solve 1 -hpddm_krylov_method gcrodr -hpddm_recycle 4 -hpddm_gmres_restart 10 -hpddm_recycle_same_system 1"),
    ("p40_gcrodr_target_lm", 4, 1, "-Nx 40 -Ny 40 -second_solve 1 -hpddm_krylov_method gcrodr -hpddm_recycle 4 -hpddm_gmres_restart 10 -hpddm_recycle_target LM"),
    ("p40_gcrodr_cycle_end", 4, 1, "-Nx 40 -Ny 40 -second_solve 1 -hpddm_krylov_method gcrodr -hpddm_recycle 4 -hpddm_gmres_restart 8"),
    # Block GCRO-DR with right-hand-side deflation (include/HPDDM_GCRODR.hpp:545-600): blocks of 3 of the 4 right-hand sides, three restarts
    ("p40_bgcrodr_rhs_deflation_mu4", 4, 4, "-Nx 40 -Ny 40 -dependent_rhs 1 -hpddm_krylov_method bgcrodr -hpddm_recycle 2 -hpddm_gmres_restart 6 -hpddm_deflation_tol 1e-6"),
    ("z_p30_bgcrodr_rhs_deflation_mu4", 4, 4, "-Nx 30 -Ny 30 -dependent_rhs 1 -complex_shift_re -10 -complex_shift_im 2 -hpddm_krylov_method bgcrodr -hpddm_recycle 2 -hpddm_gmres_restart 6 -hpddm_deflation_tol 1e-6"),
    ("p40_bgcrodr_two_solves_mu2", 4, 2, "-Nx 40 -Ny 40 -second_solve 1 -hpddm_krylov_method bgcrodr -hpddm_recycle 3 -hpddm_gmres_restart 8"),
    ("p30_6ranks_bgcrodr_left_deflated_mu3", 6, 3, "-Nx 30 -Ny 30 -overlap 2 -second_solve 1 -hpddm_krylov_method bgcrodr -hpddm_recycle 2 -hpddm_gmres_restart 8 -hpddm_variant left -hpddm_schwarz_coarse_correction deflated -hpddm_geneo_nu=0"),
    ("p30_6ranks_gcrodr_left_deflated_mu2", 6, 2, "-Nx 30 -Ny 30 -overlap 2 -second_solve 1 -hpddm_krylov_method gcrodr -hpddm_recycle 3 -hpddm_gmres_restart 8 -hpddm_variant left -hpddm_schwarz_coarse_correction deflated -hpddm_geneo_nu=0"),
    ("p40_richardson_mu2", 4, 2, "-Nx 40 -Ny 40 -hpddm_krylov_method richardson -hpddm_max_it 15 -hpddm_richardson_damping_factor 0.7"),
    ("p40_none_deflated_mu2", 4, 2, "-Nx 40 -Ny 40 -hpddm_krylov_method none -hpddm_schwarz_coarse_correction deflated -hpddm_geneo_nu=0"),
    ("p40_bgmres_mgs_qrmgs_mu3", 4, 3, "-Nx 40 -Ny 40 -hpddm_krylov_method bgmres -hpddm_orthogonalization mgs -hpddm_qr mgs -hpddm_gmres_restart=8"),
    ("p40_bgmres_qrcgs_mu3", 4, 3, "-Nx 40 -Ny 40 -hpddm_krylov_method bgmres -hpddm_qr cgs -hpddm_gmres_restart=8"),
    ("p40_cg_asm", 4, 2, "-Nx 40 -Ny 40 -hpddm_krylov_method cg -hpddm_schwarz_method asm"),
    # config 1 of BASELINE.json (45 iterations)
    ("c1_p200_onelevel", 4, 1, "-Nx 200 -Ny 200"),
]


def parse_dump(path):
    out = {}
    with open(path) as fh:
        lines = fh.read().split("\n")
    i = 0
    while i < len(lines):
        ln = lines[i]
        if ln.startswith("@"):
            name, kind, cnt = ln[1:].split()
            cnt = int(cnt)
            vals = lines[i + 1 : i + 1 + cnt]
            if kind == "z":   # complex K: "re im" pairs
                pairs = np.array([v.split() for v in vals], dtype=np.float64).reshape(-1, 2)
                out[name] = pairs[:, 0] + 1j * pairs[:, 1]
            else:
                out[name] = np.array(vals, dtype=np.float64 if kind == "f" else np.int32)
            i += 1 + cnt
        else:
            i += 1
    return out


def run_case(name, ranks, mu, opts, tmp):
    env = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL")
    cmd = ["/opt/conda/bin/mpiexec", "-n", str(ranks), os.path.join(REF, "ref_harness_z" if name.startswith("z_") else "ref_harness"), "-out", tmp, "-case", name,
           "-mu", str(mu), "-hpddm_verbosity=3"] + opts.split()
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1800)
    if res.returncode != 0:
        print(res.stdout[-2000:], res.stderr[-2000:])
        raise SystemExit(f"{name}: harness failed")
    hist = []
    for m in re.finditer(r"^(?:B?GMRES|B?CG|BFBCG|B?GCRODR):\s+(\d+)\s+(\S+)\s+(\S+)\s+(\S+)\s+<", res.stdout, re.M):
        hist.append((int(m.group(1)), float(m.group(2)), float(m.group(3)), float(m.group(4))))
    data = {"ranks": np.int32(ranks), "mu": np.int32(mu), "options": np.array(opts),
            "history": np.array(hist, dtype=np.float64).reshape(-1, 4)}
    for r in range(ranks):
        dump = parse_dump(os.path.join(tmp, f"{name}_r{r}.txt"))
        nmap = len(dump["neighbors"])
        for key, val in dump.items():
            data[f"{key}_r{r}"] = val
        assert nmap == sum(1 for k in dump if k.startswith("map_"))
    its = {int(data[f"iterations_r{r}"][0]) for r in range(ranks)}
    assert len(its) == 1
    print(f"{name}: ranks={ranks} mu={mu} it={its.pop()} hist={len(hist)} "
          f"res={data['residual_r0'][1]:.6e}/{data['residual_r0'][0]:.6e}")
    return data


def main():
    os.makedirs(OUT, exist_ok=True)
    only = set(sys.argv[1:])
    with tempfile.TemporaryDirectory() as tmp:
        for name, ranks, mu, opts in CASES:
            if only and name not in only:
                continue
            data = run_case(name, ranks, mu, opts, tmp)
            if name.startswith("c1_"):
                # config 1 is 4 x 10201 dofs: the matrices are regenerated by our generator; keep only what pins
                # the result (d, f, apply/solve outputs would be 10 MB as text) -> sol, history, residual, iterations
                keep = ("ranks", "mu", "options", "history")
                data = {k: v for k, v in data.items()
                        if k in keep or k.split("_r")[0] in ("iterations", "residual", "sol", "meta", "apply_out")}
            np.savez_compressed(os.path.join(OUT, name + ".npz"), **data)


if __name__ == "__main__":
    main()
