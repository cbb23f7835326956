"""debug helper: per-operation errors of the complex fixtures on the GPU (prints, does not assert)"""
import sys
sys.path.insert(0, "tests")
sys.path.insert(0, ".")
import numpy as np
import golden_util as gu
from hpddm_amd import hpddm


def err(a, b):
    sc = max(np.abs(np.ravel(x)).max() for x in b)
    return max(np.abs(np.ravel(x) - np.ravel(y)).max() for x, y in zip(a, b)) / sc


for name in gu.COMPLEX_CASES + gu.COMPLEX_BGMRES_CASES:
    g = gu.load(name)
    subs = gu.subdomains(g)
    opt = gu.options(g)
    A, d = hpddm.schwarz_from_subdomains(subs, options=gu.hpddm_args(g))
    if opt["correction"]:
        for s, Z in enumerate(gu.deflation_vectors(g, subs)):
            A.set_vectors(s, Z)
        if any(len(sd["neighbors"]) != len(subs) - 1 for sd in subs):
            A.set_option("hip_coarse_transpose", 1)
        A.build_coarse_operator()
    A.call_numfact()
    f = gu.vecs(g, "f")
    print(name, "d", max(np.abs(d[r] - g[f"d_r{r}"]).max() for r in range(len(subs))), flush=True)
    for what, fn in [("exchange", A.exchange), ("gmv", A.gmv), ("solve", A.local_solve), ("apply", A.apply)] + ([("deflation", A.deflation)] if opt["correction"] else []):
        print("   %-10s %.2e" % (what, err(fn(f), gu.vecs(g, what + "_out"))), flush=True)
    it, sol, hist = A.solve(f, history=True)
    ref = g["history"]
    m = min(len(hist), len(ref))
    print("   it", it, int(g["iterations_r0"][0]), "hist rel %.2e" % np.max(np.abs(hist[:m] - ref[:m, 1]) / ref[:m, 1]), "sol %.2e" % err(sol, gu.vecs(g, "sol")), flush=True)
    A.destroy()
