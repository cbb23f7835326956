#!/usr/bin/env python3
"""Counterpart of the reference's examples/solver.py:32-50 on the HIP local solver: parse a dumped matrix, factorise it
(subdomainNumfact), solve one random right-hand side (subdomainSolve) and check ||A x - f|| / ||f|| <= 1e-8.

    python examples/solver.py tests/golden/dump/out_0_4.txt
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hpddm_amd import hpddm  # noqa: E402
from hpddm_amd.matrix_io import csrmv, read_matrix  # noqa: E402


def main(path, seed=None):
    mat = read_matrix(path)
    n = mat["n"]
    S = hpddm.Subdomain()
    S.numfact(n, mat["ia"], mat["ja"], mat["a"], sym=mat["sym"])
    f = np.random.default_rng(seed).random(n)
    sol = S.solve(f)
    nrmb = np.linalg.norm(f)
    nrmAx = np.linalg.norm(csrmv(mat, sol) - f)
    print(" --- residual = {:e} / {:e}".format(nrmAx, nrmb))
    return 1 if (nrmb < 1.0e-12 or nrmAx / nrmb > 1.0e-8) else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
