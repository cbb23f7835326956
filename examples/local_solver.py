#!/usr/bin/env python3
"""Counterpart of the reference's benchmark/local_solver.cpp:92-127 protocol on the HIP local solver: a dumped matrix,
right-hand sides of ones, `warm_up` untimed runs, then `trials` rows of timings: numfact (unless --solve-phase-only) and
one solve per nu = 1, 2, 4, ..., rhs (host arrays in and out, as Solver::solve is called by the reference; the last
column set, prefixed '|', is the device-resident sweep alone).

    python examples/local_solver.py tests/golden/dump/out_0_4.txt --rhs 8 --solve-phase-only
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hpddm_amd import hpddm  # noqa: E402
from hpddm_amd.matrix_io import read_matrix  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("matrix")
    ap.add_argument("--warm-up", type=int, default=1)
    ap.add_argument("--trials", type=int, default=3)
    ap.add_argument("--rhs", type=int, default=1)
    ap.add_argument("--solve-phase-only", action="store_true")
    ap.add_argument("--spd", action="store_true", help="-hpddm_operator_spd")
    args = ap.parse_args()
    mat = read_matrix(args.matrix)
    n = mat["n"]
    rhs = np.ones((n, args.rhs), order="F")
    nus = []
    nu = args.rhs
    while nu >= 1:
        nus.insert(0, nu)
        nu //= 2

    def numfact(S):
        S.numfact(n, mat["ia"], mat["ja"], mat["a"], sym=mat["sym"], spd=args.spd)

    S = hpddm.Subdomain()
    if args.solve_phase_only:
        numfact(S)
    for _ in range(args.warm_up):
        if not args.solve_phase_only:
            numfact(S)
        S.solve(rhs)
    for _ in range(args.trials):
        row = []
        if not args.solve_phase_only:
            t = time.perf_counter()
            numfact(S)  # same pattern: the analysis is reused, like MUMPS job=2
            row.append(time.perf_counter() - t)
        for nu in nus:
            b = np.asfortranarray(rhs[:, :nu])
            t = time.perf_counter()
            S.solve(b)
            row.append(time.perf_counter() - t)
        dev = [S.time_solve(mu=nu, warmup=1, reps=5) for nu in nus]
        print("\t".join(f"{v:10.5e}" for v in row) + "\t|\t" + "\t".join(f"{v:10.5e}" for v in dev))
    return 0


if __name__ == "__main__":
    sys.exit(main())
