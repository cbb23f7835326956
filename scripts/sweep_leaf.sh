#!/bin/bash
# developer aid: nested-dissection leaf size against the SpTRSV rate (one factorisation per setting). usage: sweep_leaf.sh GRID LEAF...
cd "$(dirname "$0")/.." || exit 1
grid=$1; shift
for leaf in "$@"; do
  for ch in 1 0; do
    HPDDM_HIP_CHAINS=$ch python - <<PY
import os, sys, time
sys.path.insert(0, ".")
from hpddm_amd import hpddm
from hpddm_amd.generate import generate3d
subs = generate3d($grid, 8, overlap=1, sym=True, rhs="smooth")
A, d = hpddm.schwarz_from_subdomains(subs, options="-hpddm_operator_spd -hpddm_leaf_size $leaf")
t0 = time.time(); A.call_numfact(); ts = time.time() - t0
st = A.stats()
t = A.time("solve", mu=1, warmup=2, reps=10)
b = 2.0 * st["nnz_L"] * 8 + 4 * st["n"] * 8
print(f"leaf $leaf chains $ch: numfact {ts:.1f} s nnz(L) {st['nnz_L']:.4g} stored {st['stored']:.4g} levels {int(st['levels'])} launches {int(st['launches'])} sptrsv {t*1e3:.3f} ms frac {b/t/8e12:.4f}", flush=True)
PY
  done
done
