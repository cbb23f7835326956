#!/usr/bin/env python3
"""developer aid: set-up time of ONE subdomain (N^3 7-point Laplacian, Cholesky; analysis + numeric factorisation, upper levels on
the device) under the host thread cap of HPDDM_HIP_NUM_THREADS -- what a rank of an 8-GPU job gets of a node's CPU quota.
usage: HPDDM_HIP_NUM_THREADS=2 time_setup_threads.py [N=129]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hpddm_amd import hpddm  # noqa: E402
from hpddm_amd.generate import generate3d  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 129
sd = generate3d(N - 1, 1, 0, sym=True)[0]
S = hpddm.Subdomain()
t0 = time.time()
S.numfact(sd["n"], sd["ia"], sd["ja"], sd["a"], sym=True, spd=True)
t1 = tfirst = time.time()
info = S.info()
best = 1e30
for _ in range(int(os.environ.get("REFACT", "1"))):
    t1 = time.time()
    S.numfact(sd["n"], sd["ia"], sd["ja"], sd["a"], sym=True, spd=True)   # same pattern: numerical phase only
    best = min(best, time.time() - t1)
t2 = t1 + best
print(f"threads {os.environ.get('HPDDM_HIP_NUM_THREADS', 'default')}: n {sd['n']}, first numfact {tfirst - t0:.2f} s (ordering {info['t_order']:.2f}, symbolic {info['t_symbolic']:.2f}, "
      f"numeric {info['t_numeric']:.2f}, upload {info['t_upload']:.2f}), refactorisation {t2 - t1:.2f} s", flush=True)
