"""ctypes wrapper of oracle/sptrsv_oracle.c (CPU substitution on the plain supernodal factor) -- TEST INFRASTRUCTURE ONLY."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


class _Factor(ctypes.Structure):
    _fields_ = [("n", ctypes.c_longlong), ("nblk", ctypes.c_longlong), ("kind", ctypes.c_int)] + \
               [(k, ctypes.c_void_p) for k in ("perm", "blk_ptr", "ldw", "f_off", "row_ptr", "rows", "L", "U", "dinv")]


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle_sptrsv.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
        _lib = ctypes.CDLL(path)
        _lib.oracle_sptrsv_batch.restype = ctypes.c_double
        _lib.oracle_sptrsv_batch_z.restype = ctypes.c_double
        _lib.oracle_sptrsv_batch_levels.restype = ctypes.c_double
        _lib.oracle_sptrsv_batch_teams.restype = ctypes.c_double
        assert _lib.oracle_factor_sizeof() == ctypes.sizeof(_Factor)
    return _lib


class PlainFactor:
    """host copy of the plain supernodal factor exported by the product (needs numfact with keep_plain=1)"""

    def __init__(self, sub):
        info = sub.info()
        self.n, self.kind = info["n"], info["kind"]
        self.complex = bool(getattr(sub, "complex", False))   # the plain pools then hold (re, im) pairs; n, offsets, leading dimensions count complex scalars
        self.arr = {k: sub.export(k) for k in ("perm", "blk_ptr", "ldw", "f_off", "row_ptr", "rows", "height")}
        view = getattr(sub, "export_view", None) or sub.export   # no copy of a multi-GB factor when the binding offers a view
        self.sub = sub                                            # the views point into the solver's storage
        self.arr["L"] = view("Lplain")
        self.arr["U"] = view("Uplain") if self.kind == 2 else np.zeros(1)
        self.arr["dinv"] = view("dinv") if self.kind == 1 else np.zeros(1)
        assert self.arr["L"].size > 0, "numfact was not run with keep_plain=1"
        self.nnz_bytes = self.arr["L"].nbytes

    def struct(self):
        f = _Factor()
        f.n, f.nblk, f.kind = self.n, len(self.arr["blk_ptr"]) - 1, self.kind
        for k in ("perm", "blk_ptr", "ldw", "f_off", "row_ptr", "rows", "L", "U", "dinv"):
            setattr(f, k, self.arr[k].ctypes.data)
        return f

    def solve(self, b):
        b = np.asfortranarray(b, dtype=np.float64)
        x = np.empty_like(b, order="F")
        mu = 1 if b.ndim == 1 else b.shape[1]
        a = self.arr
        P = ctypes.c_void_p
        lib().oracle_sptrsv(ctypes.c_longlong(self.n), ctypes.c_longlong(len(a["blk_ptr"]) - 1), self.kind, P(a["perm"].ctypes.data), P(a["blk_ptr"].ctypes.data),
                            P(a["ldw"].ctypes.data), P(a["f_off"].ctypes.data), P(a["row_ptr"].ctypes.data), P(a["rows"].ctypes.data), P(a["L"].ctypes.data),
                            P(a["U"].ctypes.data), P(a["dinv"].ctypes.data), P(b.ctypes.data), P(x.ctypes.data), mu)
        return x


def time_batch(factors, bs, reps, threads):
    """wall seconds for `reps` x (one solve of every subdomain), `threads` subdomains at a time"""
    n = len(factors)
    arr = (_Factor * n)(*[f.struct() for f in factors])
    bs = [np.asfortranarray(b, dtype=np.float64) for b in bs]
    xs = [np.empty_like(b) for b in bs]
    mu = 1 if bs[0].ndim == 1 else bs[0].shape[1]
    bp = (ctypes.c_void_p * n)(*[b.ctypes.data for b in bs])
    xp = (ctypes.c_void_p * n)(*[x.ctypes.data for x in xs])
    sec = lib().oracle_sptrsv_batch(n, arr, bp, xp, mu, reps, threads)
    return sec, xs


def time_batch_z(factors, bs, reps, threads):
    """the same for complex factors (K = std::complex<double>: plain L D L^T with plain transposes, or LU): bs[s] complex, (n,) or (n, mu)"""
    n = len(factors)
    assert all(getattr(f, "complex", False) for f in factors)
    arr = (_Factor * n)(*[f.struct() for f in factors])
    bs = [np.asfortranarray(b, dtype=np.complex128) for b in bs]
    xs = [np.empty_like(b) for b in bs]
    mu = 1 if bs[0].ndim == 1 else bs[0].shape[1]
    bp = (ctypes.c_void_p * n)(*[b.ctypes.data for b in bs])
    xp = (ctypes.c_void_p * n)(*[x.ctypes.data for x in xs])
    sec = lib().oracle_sptrsv_batch_z(n, arr, bp, xp, mu, reps, threads)
    return sec, xs


def time_batch_teams(factors, bs, reps, team):
    """the same with a team of `team` threads per subdomain (nested OpenMP: the row loops of the large supernodes shared)"""
    n = len(factors)
    arr = (_Factor * n)(*[f.struct() for f in factors])
    bs = [np.ascontiguousarray(b, dtype=np.float64) for b in bs]
    assert all(b.ndim == 1 for b in bs)
    xs = [np.empty_like(b) for b in bs]
    bp = (ctypes.c_void_p * n)(*[b.ctypes.data for b in bs])
    xp = (ctypes.c_void_p * n)(*[x.ctypes.data for x in xs])
    sec = lib().oracle_sptrsv_batch_teams(n, arr, bp, xp, reps, team)
    return sec, xs


def time_batch_levels(factors, bs, reps, threads):
    """the same on `threads` cores: level-scheduled over the assembly tree, all subdomains at once (one right-hand side)"""
    n = len(factors)
    arr = (_Factor * n)(*[f.struct() for f in factors])
    bs = [np.ascontiguousarray(b, dtype=np.float64) for b in bs]
    assert all(b.ndim == 1 for b in bs)
    xs = [np.empty_like(b) for b in bs]
    bp = (ctypes.c_void_p * n)(*[b.ctypes.data for b in bs])
    xp = (ctypes.c_void_p * n)(*[x.ctypes.data for x in xs])
    hp = (ctypes.c_void_p * n)(*[f.arr["height"].ctypes.data for f in factors])
    sec = lib().oracle_sptrsv_batch_levels(n, arr, hp, bp, xp, reps, threads)
    return sec, xs
