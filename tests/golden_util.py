"""Loading the golden fixtures dumped from the compiled reference (oracle/make_golden.py)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

SMALL_CASES = ["p40_onelevel", "p40_onelevel_mu3", "p40_sym_spd", "p40_sym_ldlt", "p40_overlap2", "p40_deflated",
               "p40_deflated_mu2_ov2", "p40_additive", "p40_balanced", "p36x60_9ranks", "p30_6ranks_mgs_left"]


def load(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: g[k] for k in g.files}


def options(g):
    """reference command line of the case -> dict of the options the path reads"""
    toks = str(g["options"]).split()
    out = {"tol": 1e-6, "max_it": 100, "restart": 40, "variant": "right", "ortho": "cgs", "correction": None, "spd": False, "method": "ras",
           "deflation_tol": -1.0, "recycle_target": "SM", "qr": "cholqr"}
    i = 0
    while i < len(toks):
        t = toks[i]
        if t.startswith("-hpddm_"):
            key = t[7:]
            val = None
            if "=" in key:
                key, val = key.split("=", 1)
            elif i + 1 < len(toks) and not toks[i + 1].startswith("-"):
                i += 1
                val = toks[i]
            if key == "gmres_restart":
                out["restart"] = int(val)
            elif key == "max_it":
                out["max_it"] = int(val)
            elif key == "tol":
                out["tol"] = float(val)
            elif key == "variant":
                out["variant"] = val
            elif key == "orthogonalization":
                out["ortho"] = val
            elif key == "schwarz_coarse_correction":
                out["correction"] = val
            elif key == "operator_spd":
                out["spd"] = True
            elif key == "schwarz_method":
                out["method"] = val
            elif key == "qr":
                out["qr"] = val
            elif key == "recycle_target":
                out["recycle_target"] = val
            elif key == "deflation_tol":
                out["deflation_tol"] = float(val)
        i += 1
    return out


OPTIMIZED_CASES = ["p40_oras_og", "p40_soras_os_sym", "p40_soras_os_deflated", "p40_osm_og"]
PENALIZED_CASES = ["p40_penalized_mu2", "p40_penalized_sym_left", "p40_penalized_left_mu2_ov2"]


def optimized_matrices(g, subs):
    """the optimised local matrices the harness handed to callNumfact(A) (same pattern as the subdomain matrices, values a_opt)"""
    out = []
    for r, s in enumerate(subs):
        t = dict(s)
        t["a"] = g[f"a_opt_r{r}"]
        out.append(t)
    return out


def hpddm_args(g):
    """the -hpddm_* part of the reference command line (passed verbatim to HpddmHipSchwarzOptionParse)"""
    toks = str(g["options"]).split()
    out, i = [], 0
    while i < len(toks):
        if toks[i].startswith("-hpddm_"):
            out.append(toks[i])
            if "=" not in toks[i] and i + 1 < len(toks) and not toks[i + 1].startswith("-"):
                i += 1
                out.append(toks[i])
        i += 1
    return " ".join(out)


def subdomains(g):
    """per-rank dicts (same keys as hpddm_amd.generate) rebuilt from a fixture that carries its matrices"""
    subs = []
    for r in range(int(g["ranks"])):
        meta = g[f"meta_r{r}"]
        nb = g[f"neighbors_in_r{r}"]
        subs.append(dict(n=int(meta[2]), ia=g[f"ia_r{r}"], ja=g[f"ja_r{r}"], a=g[f"a_r{r}"], sym=bool(meta[4]), numbering="C",
                         neighbors=nb, connectivity=[g[f"mapping_in_{k}_r{r}"] for k in range(len(nb))], d=g[f"d_in_r{r}"],
                         f=g[f"f_r{r}"]))
    return subs


COMPLEX_CASES = ["z_p30_gmres_mu2", "z_p30_gmres_left_deflated", "z_p30_6ranks_deflated_nu3", "z_p30_oras_og_mu2", "z_p30_soras_os_deflated",
                 "z_p30_additive_mu2", "z_p30_fgmres_restart8_mu2", "z_p36x60_9ranks_mu2"]          # K = std::complex<double>, GMRES
COMPLEX_BGMRES_CASES = ["z_p30_6ranks_bgmres_mu3_balanced", "z_p30_bgmres_mu8", "z_p30_bgmres_rhs_deflation_mu4", "z_p30_fbgmres_mu3"]
MULTI_VECTOR_CASES = ["p30_6ranks_deflated_nu3"]   # three deflation vectors per subdomain, non-symmetric local matrices


def deflation_vectors(g, subs):
    """the vectors the harness handed to setVectors: the constant one (examples/schwarz.cpp:115-121), or the dumped `ev`
    block (n x nu, column-major) of the fixtures made with -deflation_nu"""
    if "ev_r0" in g:
        return [g[f"ev_r{r}"].reshape(-1, sd["n"]).T.copy() for r, sd in enumerate(subs)]
    return [np.ones((sd["n"], 1)) for sd in subs]


def vec(g, key, r, mu):
    n = int(g[f"meta_r{r}"][2])
    v = g[f"{key}_r{r}"]
    return v.copy() if mu == 1 else v.reshape(n, mu, order="F").copy(order="F")


def vecs(g, key):
    mu = int(g["mu"])
    return [vec(g, key, r, mu) for r in range(int(g["ranks"]))]
