import time, numpy as np, sys
sys.path.insert(0,'/root/repo')
from hpddm_amd import hpddm
from hpddm_amd.generate import generate3d
subs = generate3d(14, 8, 2, sym=True, rhs="smooth", neumann=True)
A, d = hpddm.schwarz_from_subdomains(subs, options="-hpddm_operator_spd -hpddm_schwarz_coarse_correction deflated -hpddm_geneo_nu 6 -hpddm_verbosity 2")
for s, sd in enumerate(subs[:2]):
    t=time.time(); lam = A.solve_gevp(s, sd["n"], sd["ia"], sd["ja"], sd["a_neumann"], sd["sym"]); print(time.time()-t, lam)
