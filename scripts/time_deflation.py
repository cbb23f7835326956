#!/usr/bin/env python3
"""developer aid: the deflation panel and the GMV of configs[2] (8 subdomains of 129^3, 20 polynomial vectors each) timed without
the factorisation -- the coarse correction needs Z, d, A and the coarse operator only.
usage: time_deflation.py [grid=256 | helmholtz] ["-hpddm_opt v ..." ...]      (helmholtz: the complex share of configs[4], 8 right-hand sides)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hpddm_amd import hpddm  # noqa: E402
from hpddm_amd.generate import generate3d  # noqa: E402

helm = len(sys.argv) > 1 and sys.argv[1] == "helmholtz"
N = 256 if helm or len(sys.argv) < 2 else int(sys.argv[1])
cfgs = sys.argv[2:] or [""]
nu = 20
if helm:
    from hpddm_amd.generate import generate_helmholtz3d
    subs = generate_helmholtz3d((64, 64, 128), 8, grid=(2, 2, 2))
    A, d = hpddm.schwarz_from_subdomains(subs, options="-hpddm_schwarz_coarse_correction deflated", multiplicity=False)
    for s, sd in enumerate(subs):
        t = np.arange(sd["n"], dtype=np.float64)
        A.set_vectors(s, np.stack([np.ones(sd["n"], dtype=np.complex128), np.exp(0.21j * t), np.exp(-0.13j * t + 0.4j * s)], axis=1))
    A.build_coarse_operator()
    for cfg in cfgs:
        if cfg:
            A.option_parse(cfg)
        for mu in (1, 8):
            print(f"[{cfg}] helmholtz mu {mu}: deflation {A.time('deflation', mu, 3, 50) * 1e3:.3f} ms, gmv {A.time('gmv', mu, 3, 50) * 1e3:.3f} ms, exchange {A.time('exchange', mu, 3, 50) * 1e3:.3f} ms, halo in place {A.time('halo', mu, 3, 50) * 1e3:.3f} ms", flush=True)
    sys.exit(0)
subs = generate3d(N, 8, overlap=1, sym=True, rhs="smooth")
A, d = hpddm.schwarz_from_subdomains(subs, options="-hpddm_operator_spd -hpddm_schwarz_coarse_correction deflated")
expo = [(a, b, c) for deg in range(8) for a in range(deg + 1) for b in range(deg + 1 - a) for c in [deg - a - b]][:nu]
for s, sd in enumerate(subs):
    i0, i1, j0, j1, k0, k1 = sd["box"]
    z, y, x = np.meshgrid(np.linspace(-1, 1, k1 - k0), np.linspace(-1, 1, j1 - j0), np.linspace(-1, 1, i1 - i0), indexing="ij")
    A.set_vectors(s, np.stack([(x ** a * y ** b * z ** c).ravel() for a, b, c in expo], axis=1))
t0 = time.time()
A.build_coarse_operator()
print(f"coarse operator {time.time() - t0:.2f} s, n = {int(A.stats()['n'])}", flush=True)
zbytes = 2.0 * nu * A.stats()["n"] * 8.0
for cfg in cfgs:
    if cfg:
        A.option_parse(cfg)
    for mu in [int(v) for v in os.environ.get("MUS", "1,2").split(",")]:
        td, tg = A.time("deflation", mu, 3, 30), A.time("gmv", mu, 3, 30)
        te, th = A.time("exchange", mu, 3, 30), A.time("halo", mu, 3, 30)
        pb = zbytes + 3.0 * A.stats()["n"] * mu * 8.0
        print(f"[{cfg}] mu {mu}: deflation {td * 1e3:.3f} ms ({zbytes / td / 1e9:.0f} GB/s on 2 x Z, {pb / td / 1e9:.0f} GB/s on the panel bytes of SURVEY 8(d)), gmv {tg * 1e3:.3f} ms, exchange {te * 1e3:.3f} ms, halo in place {th * 1e3:.3f} ms", flush=True)
