"""Python host side of the MI355X RAS path, mirroring the reference's own ctypes binding ``interface/hpddm.py``
(names ``subdomainNumfact``/``subdomainSolve``/``schwarzCreate``/... , hpddm.py:175-275) on top of the C ABI of
``include/hpddm_hip.h``.  numpy arrays are host memory; device-resident variants take raw device pointers
(e.g. ``torch.Tensor.data_ptr()``).

Difference with the reference that the hardware imposes: ONE process drives ALL subdomains of a GPU, so a
:class:`Schwarz` object is created for ``nsub`` subdomains and multi-vectors are lists of per-subdomain arrays
(each ``(n_s,)`` or ``(n_s, mu)`` Fortran-ordered, exactly what one MPI rank of the reference holds).
"""
import ctypes
import os

import numpy as np

from . import _lib
from ._lib import HpddmHipError, check


def _dptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _as_f(a, dtype=np.float64):
    return np.asfortranarray(a, dtype=dtype)


def device_count():
    return _lib.load().HpddmHipDeviceCount()


def rccl_unique_id():
    """ncclGetUniqueId through the library (HpddmHipRcclGetUniqueId): 128 bytes, to be made on one rank and handed to the others"""
    buf = ctypes.create_string_buffer(128)
    check(_lib.load().HpddmHipRcclGetUniqueId(buf))
    return buf.raw


def rccl_halo_probe(A, unique_id, send, mu, red_sum=None, red_max=None):
    """HpddmHipRcclHaloProbe on host arrays (no device; HPDDM_HIP_RCCL_LIB must name a host-side double of librccl): returns the receive
    buffer of one exchange of ``A``'s peer layout; ``red_sum`` / ``red_max`` (same length) are reduced in place over the ranks"""
    send = np.ascontiguousarray(send, dtype=np.float64)
    recv = np.zeros_like(send)
    n = 0 if red_sum is None else len(red_sum)
    check(_lib.load().HpddmHipRcclHaloProbe(A._h, ctypes.create_string_buffer(unique_id, 128), _dptr(send), _dptr(recv), int(mu), _dptr(red_sum) if n else None,
                                            _dptr(red_max) if n else None, n))
    return recv


def rccl_self_test():
    """one-rank check of the RCCL transport on the library stream (HpddmHipRcclSelfTest)"""
    check(_lib.load().HpddmHipRcclSelfTest())


def require_device():
    """The product has no CPU path: fail loudly when no MI355X is visible."""
    if device_count() < 1:
        raise HpddmHipError("no HIP device visible: libhpddm_hip.so needs an MI355X (there is no CPU fallback)")


# ---------------------------------------------------------------------------------------------------------------------
class Subdomain:
    """Local solver: ``SUBDOMAIN<K>`` of the reference (Solver concept, include/HPDDM_MUMPS.hpp:206-318)."""

    def __init__(self, **options):
        self._lib = _lib.load()
        self._h = ctypes.c_void_p()
        self._owned = True
        for k, v in options.items():
            check(self._lib.HpddmHipSubdomainSetOption(ctypes.byref(self._h), k.encode(), float(v)))
        self.n = 0

    @classmethod
    def _borrow(cls, handle):
        self = cls.__new__(cls)
        self._lib = _lib.load()
        self._h = ctypes.c_void_p(handle)
        self._owned = False
        self.n = int(self.info()["n"])
        return self

    def numfact(self, n, ia, ja, a, sym=False, numbering="C", spd=False):
        """subdomainNumfact (interface/hpddm.py:185, HpddmSubdomainNumfact interface/HPDDM.h:88).  A complex ``a`` takes
        the complex128 entry point (the reference built with a complex scalar type)."""
        ia = np.ascontiguousarray(ia, dtype=np.int32)
        ja = np.ascontiguousarray(ja, dtype=np.int32)
        self.complex = np.iscomplexobj(a)
        if self.complex:
            a = np.ascontiguousarray(a, dtype=np.complex128)
            fn = self._lib.HpddmHipSubdomainNumfactZ
        else:
            a = np.ascontiguousarray(a, dtype=np.float64)
            fn = self._lib.HpddmHipSubdomainNumfact
        check(fn(ctypes.byref(self._h), int(n), _dptr(ia), _dptr(ja), _dptr(a), int(bool(sym)), numbering.encode(), int(bool(spd))))
        self.n = int(n)

    def solve(self, f, sol=None):
        """subdomainSolve (interface/hpddm.py:191): sol = A^{-1} f; f is (n,) or (n, mu) Fortran-ordered."""
        cplx = getattr(self, "complex", False)
        f = _as_f(f, np.complex128 if cplx else np.float64)
        mu = 1 if f.ndim == 1 else f.shape[1]
        if sol is None:
            sol = np.empty_like(f, order="F")
        assert sol.flags.f_contiguous and sol.shape == f.shape and sol.dtype == f.dtype
        check((self._lib.HpddmHipSubdomainSolveZ if cplx else self._lib.HpddmHipSubdomainSolve)(self._h, _dptr(f), _dptr(sol), mu))
        return sol

    def inertia(self):
        """Solver::inertia: negative pivots of the last factorisation (-1: the factor is an LU one, or complex)"""
        return int(self._lib.HpddmHipSubdomainInertia(self._h))

    def refine_steps(self):
        """steps of iterative refinement every solve takes (0 unless the probe solve of numfact found a factor that is not backward stable
        by itself but whose error contracts)"""
        return int(self._lib.HpddmHipSubdomainRefineSteps(self._h))

    def solve_device(self, b_ptr, x_ptr, mu=1):
        check(self._lib.HpddmHipSubdomainSolveDevice(self._h, ctypes.c_void_p(b_ptr), ctypes.c_void_p(x_ptr), mu))

    def info(self):
        info = np.zeros(12, dtype=np.int64)
        times = np.zeros(4)
        check(self._lib.HpddmHipSubdomainInfo(self._h, _dptr(info), _dptr(times)))
        keys = ("n", "supernodes", "levels", "nnz_L", "stored", "pool", "update_pool", "kind", "launches", "flops", "plain_export_us", "bushes")
        out = {k: int(v) for k, v in zip(keys, info)}
        out.update(t_order=times[0], t_symbolic=times[1], t_numeric=times[2], t_upload=times[3])
        return out

    def export(self, which):
        ints = which not in ("F", "G", "dinv", "Lplain", "Uplain")
        cnt = self._lib.HpddmHipSubdomainExport(self._h, which.encode(), None, 0)
        if cnt < 0:
            raise HpddmHipError(self._lib.HpddmHipLastError().decode())
        out = np.zeros(cnt, dtype=np.int64 if ints else np.float64)
        if cnt:
            check(self._lib.HpddmHipSubdomainExport(self._h, which.encode(), _dptr(out), cnt))
        return out

    def export_view(self, which):
        """the double arrays of :meth:`export` without a copy (numpy view of the solver's own storage; keep the solver alive)"""
        cnt = ctypes.c_longlong()
        ptr = self._lib.HpddmHipSubdomainExportView(self._h, which.encode(), ctypes.byref(cnt))
        if not ptr:
            raise HpddmHipError(self._lib.HpddmHipLastError().decode())
        if cnt.value == 0:
            return np.zeros(0)
        return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_double)), shape=(cnt.value,))

    def time_solve(self, mu=1, warmup=2, reps=10):
        sec = ctypes.c_double()
        check(self._lib.HpddmHipSubdomainTimeSolve(self._h, mu, warmup, reps, ctypes.byref(sec)))
        return sec.value

    def destroy(self):
        if self._owned and self._h:
            self._lib.HpddmHipSubdomainDestroy(self._h)
        self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


# reference-style free functions (interface/hpddm.py:183-199)
def subdomainNumfact(S, n, ia, ja, a, sym=False, numbering="C", spd=False):
    if S is None:
        S = Subdomain()
    S.numfact(n, ia, ja, a, sym, numbering, spd)
    return S


def subdomainSolve(S, f, sol):
    S.solve(f, sol)


def subdomainDestroy(S):
    S.destroy()


# ---------------------------------------------------------------------------------------------------------------------
class Schwarz:
    """``HPDDM::Schwarz`` for all the subdomains of one GPU (HpddmSchwarz* of interface/HPDDM.h:101-112)."""

    def __init__(self, nsub, first_global=0, nglobal=None):
        self._lib = _lib.load()
        nglobal = first_global + nsub if nglobal is None else nglobal
        self._h = self._lib.HpddmHipSchwarzCreate(nsub, first_global, nglobal)
        if not self._h:
            raise HpddmHipError(self._lib.HpddmHipLastError().decode())
        self.nsub, self.first, self.nglobal = nsub, first_global, nglobal
        self.n = [0] * nsub
        self.complex = False   # K = std::complex<double>: set by the first complex subdomain matrix

    # -- construction, same order as examples/schwarz.cpp:90-126 --
    def set_subdomain(self, s, n, ia, ja, a, sym, neighbors, connectivity, numbering="C"):
        """schwarzCreate(Mat, o, connectivity) (interface/hpddm.py:216) for local subdomain s."""
        ia = np.ascontiguousarray(ia, dtype=np.int32)
        ja = np.ascontiguousarray(ja, dtype=np.int32)
        cplx = np.iscomplexobj(a)
        if cplx != self.complex and any(self.n):
            raise HpddmHipError("real and complex subdomain matrices cannot be mixed")
        self.complex = cplx
        a = np.ascontiguousarray(a, dtype=np.complex128 if cplx else np.float64)   # complex128 = interleaved (re, im) doubles
        neighbors = np.ascontiguousarray(neighbors, dtype=np.int32)
        conn = [np.ascontiguousarray(c, dtype=np.int32) for c in connectivity]
        sizes = np.array([c.size for c in conn], dtype=np.int32)
        ptrs = (ctypes.c_void_p * max(1, len(conn)))(*[c.ctypes.data for c in conn])
        setter = self._lib.HpddmHipSchwarzSetSubdomainZ if cplx else self._lib.HpddmHipSchwarzSetSubdomain
        check(setter(self._h, s, int(n), _dptr(ia), _dptr(ja), _dptr(a), int(bool(sym)), numbering.encode(),
                                                    len(conn), _dptr(neighbors), _dptr(sizes), ctypes.cast(ptrs, ctypes.c_void_p)))
        self.n[s] = int(n)

    def multiplicity_scaling(self, d):
        """schwarzMultiplicityScaling: d is a list of per-subdomain weight arrays, overwritten by the partition of unity."""
        d = [np.ascontiguousarray(x, dtype=np.float64) for x in d]
        ptrs = (ctypes.c_void_p * self.nsub)(*[x.ctypes.data for x in d])
        check(self._lib.HpddmHipSchwarzMultiplicityScaling(self._h, ctypes.cast(ptrs, ctypes.c_void_p)))
        return d

    def initialize(self, d):
        """schwarzInitialize for every subdomain."""
        for s, x in enumerate(d):
            x = np.ascontiguousarray(x, dtype=np.float64)
            check(self._lib.HpddmHipSchwarzInitialize(self._h, s, _dptr(x)))

    def set_vectors(self, s, Z):
        """setVectors + initializeCoarseOperator (interface/hpddm.py:203-208): Z is (n_s, nu)."""
        Z = _as_f(Z, np.complex128 if self.complex else np.float64)
        Z = Z.reshape(Z.shape[0], -1, order="F")
        setter = self._lib.HpddmHipSchwarzSetVectorsZ if self.complex else self._lib.HpddmHipSchwarzSetVectors
        check(setter(self._h, s, Z.shape[1], _dptr(Z)))

    def solve_gevp(self, s, n, ia, ja, a, sym, numbering="C", B=None):
        """schwarzSolveGEVP(A, MatNeumann) (interface/hpddm.py:244): GenEO vectors of local subdomain s; returns the eigenvalues.
        B = (ia, ja, a, sym): the right-hand side matrix of Schwarz::solveGEVP(A, B) (default: scaleIntoOverlap(A)); complex operators
        take complex matrices and return complex eigenvalues (smallest modulus first)."""
        ia = np.ascontiguousarray(ia, dtype=np.int32)
        ja = np.ascontiguousarray(ja, dtype=np.int32)
        dt = np.complex128 if self.complex else np.float64
        a = np.ascontiguousarray(a, dtype=dt)
        if B is None and not self.complex:
            check(self._lib.HpddmHipSchwarzSolveGEVP(self._h, s, int(n), _dptr(ia), _dptr(ja), _dptr(a), int(bool(sym)), numbering.encode()))
        else:
            bia = bja = ba = None
            bsym = 0
            if B is not None:
                bia, bja = np.ascontiguousarray(B[0], dtype=np.int32), np.ascontiguousarray(B[1], dtype=np.int32)
                ba, bsym = np.ascontiguousarray(B[2], dtype=dt), int(bool(B[3]))
            check(self._lib.HpddmHipSchwarzSolveGEVPWith(self._h, s, int(n), _dptr(ia), _dptr(ja), _dptr(a), int(bool(sym)), numbering.encode(),
                                                         _dptr(bia) if B is not None else None, _dptr(bja) if B is not None else None,
                                                         _dptr(ba) if B is not None else None, bsym))
        if self.complex:
            k = self._lib.HpddmHipSchwarzGetEigenvaluesZ(self._h, s, None, 0)
            ev = np.zeros(max(k, 1), dtype=np.complex128)
            self._lib.HpddmHipSchwarzGetEigenvaluesZ(self._h, s, _dptr(ev), k)
            return ev[:k]
        k = self._lib.HpddmHipSchwarzGetEigenvalues(self._h, s, None, 0)
        ev = np.zeros(max(k, 1))
        self._lib.HpddmHipSchwarzGetEigenvalues(self._h, s, _dptr(ev), k)
        return ev[:k]

    def solve_gevp_all(self, mats, threads=None):
        """solve_gevp for every local subdomain (the reference's ranks each solve their own eigenproblem, side by side): `mats[s]` =
        (n, ia, ja, a, sym) or (n, ia, ja, a, sym, B); three host threads keep three subdomains in flight (HPDDM_HIP_GEVP_THREADS) -- the
        lower levels of one shifted factorisation run on the host cores while the device works on another's upper levels and on the
        block-Krylov iterations of a third.  An eigenproblem that does not find the device memory for its factor while the others hold
        theirs is solved again once they are through, alone.  Returns the list of eigenvalue arrays."""
        import threading
        out, err, again = [None] * len(mats), [], []
        it = iter(range(len(mats)))
        lock = threading.Lock()

        def one(s):
            m = mats[s]
            return self.solve_gevp(s, m[0], m[1], m[2], m[3], m[4], B=m[5] if len(m) > 5 else None)

        def work():
            while True:
                with lock:
                    s = next(it, None)
                if s is None or err:
                    return
                try:
                    out[s] = one(s)
                except _lib.HpddmHipError as e:
                    if "device memory" in str(e) and threads > 1:
                        with lock:
                            again.append(s)
                        return   # (one eigenproblem fewer in flight from here on)
                    err.append(e)
                except Exception as e:   # noqa: BLE001
                    err.append(e)
        if threads is None:
            threads = int(os.environ.get("HPDDM_HIP_GEVP_THREADS", "3"))
        threads = max(1, min(threads, len(mats)))
        ts = [threading.Thread(target=work) for _ in range(threads)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if err:
            raise err[0]
        rest = sorted(again) + list(it)   # (every worker may have left: what none of them took)
        for s in rest:
            out[s] = one(s)
        return out

    def get_vectors(self, s):
        """getVectors: the deflation vectors of local subdomain s, (n_s, nu)"""
        nu = self._lib.HpddmHipSchwarzGetVectors(self._h, s, None, 0)
        n = self.n[s]
        out = np.zeros(n * max(nu, 1), dtype=np.complex128 if self.complex else np.float64)
        if self._lib.HpddmHipSchwarzGetVectors(self._h, s, _dptr(out), out.size * (2 if self.complex else 1)) < 0:
            raise HpddmHipError(self._lib.HpddmHipLastError().decode())
        return out[:n * nu].reshape(n, nu, order="F")

    def set_optimized_matrix(self, s, n, ia, ja, a, sym, numbering="C"):
        """callNumfact(A) of the reference: optimised local matrix of subdomain s for -hpddm_schwarz_method oras / soras / osm"""
        ia = np.ascontiguousarray(ia, dtype=np.int32)
        ja = np.ascontiguousarray(ja, dtype=np.int32)
        a = np.ascontiguousarray(a, dtype=np.complex128 if self.complex else np.float64)
        setter = self._lib.HpddmHipSchwarzSetOptimizedMatrixZ if self.complex else self._lib.HpddmHipSchwarzSetOptimizedMatrix
        check(setter(self._h, s, int(n), _dptr(ia), _dptr(ja), _dptr(a), int(bool(sym)), numbering.encode()))

    def destroy_recycling(self):
        """OptionsPrefix::destroy: drop the subspace GCRO-DR recycles between successive solves"""
        check(self._lib.HpddmHipSchwarzDestroyRecycling(self._h))

    def build_coarse_operator(self):
        check(self._lib.HpddmHipSchwarzBuildCoarseOperator(self._h))

    def call_numfact(self):
        check(self._lib.HpddmHipSchwarzCallNumfact(self._h))

    def option_parse(self, args):
        """optionParse (interface/hpddm.py:124): '-hpddm_key value' strings with the reference's names."""
        check(self._lib.HpddmHipSchwarzOptionParse(self._h, args.encode()))

    def set_option(self, key, value):
        check(self._lib.HpddmHipSchwarzSetOption(self._h, key.encode(), float(value)))

    def get_option(self, key):
        return self._lib.HpddmHipSchwarzGetOption(self._h, key.encode())

    # -- several GPUs (one process per GPU): SURVEY 8(e) --
    def set_partition(self, rank, firsts):
        """rank r owns the global subdomains firsts[r] .. firsts[r+1]-1"""
        firsts = np.ascontiguousarray(firsts, dtype=np.int32)
        check(self._lib.HpddmHipSchwarzSetPartition(self._h, len(firsts) - 1, int(rank), _dptr(firsts)))

    def halo_peers(self):
        """[(peer rank, values per right-hand side, offset)] of the cross-GPU halo"""
        n = self._lib.HpddmHipSchwarzHaloPeers(self._h, 0, None, None, None)
        if n < 0:
            raise HpddmHipError(self._lib.HpddmHipLastError().decode())
        ranks, counts, offs = np.zeros(max(n, 1), dtype=np.int32), np.zeros(max(n, 1), dtype=np.int64), np.zeros(max(n, 1), dtype=np.int64)
        check(self._lib.HpddmHipSchwarzHaloPeers(self._h, n, _dptr(ranks), _dptr(counts), _dptr(offs)))
        return [(int(ranks[p]), int(counts[p]), int(offs[p])) for p in range(n)]

    def halo_export(self, which):
        cnt = self._lib.HpddmHipSchwarzHaloExport(self._h, which.encode(), None, 0)
        if cnt < 0:
            raise HpddmHipError(self._lib.HpddmHipLastError().decode())
        out = np.zeros(max(cnt, 1), dtype=np.int32)
        check(self._lib.HpddmHipSchwarzHaloExport(self._h, which.encode(), _dptr(out), cnt))
        return out[:cnt]

    def enable_rccl(self, unique_id, mu_cap=8):
        """The product transport on a multi-GPU node (one process per GPU): RCCL inside the library.  Halo = one grouped
        ncclSend / ncclRecv pair per neighbouring GPU, coarse gather and Krylov reductions = ncclAllReduce on device buffers,
        all enqueued on the library stream (HpddmHipSchwarzInitRccl).  ``unique_id``: the 128 bytes returned by
        :func:`rccl_unique_id` on ONE rank and handed to the others by the host framework (torch.distributed.broadcast_object_list,
        MPI_Bcast, a file).  Collective; call it after set_partition / set_subdomain."""
        assert len(unique_id) == 128
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        check(self._lib.HpddmHipSchwarzInitRccl(self._h, buf, int(mu_cap)))

    def enable_distributed(self, dist, device, mu_cap=8, host_staging=False):
        """Test double of :meth:`enable_rccl` (and the way a host framework that owns its communicator plugs in): registers
        callbacks on a torch.distributed process group -- grouped point-to-point send/recv per neighbouring GPU, the counterpart
        of the MPI_Isend/Irecv pairs of Subdomain::exchange, and a small all-reduce for the inner products.
        host_staging=True moves the buffers through host memory (gloo), used to test the path on a single-GPU box."""
        import torch
        peers = self.halo_peers()
        total = sum(c for _, c, _ in peers)
        self._send = torch.zeros(max(1, total * mu_cap), dtype=torch.float64, device=device)
        self._recv = torch.zeros_like(self._send)
        self._red = torch.zeros(4096, dtype=torch.float64, device="cpu" if host_staging else device)

        plans = {}  # per mu: the (send, recv, peer) views of the two buffers, built once

        def halo(ctx, mu):
            try:
                plan = plans.get(mu)
                if plan is None:
                    plan = []
                    for rank, cnt, off in peers:
                        sb, rb = self._send[off * mu:(off + cnt) * mu], self._recv[off * mu:(off + cnt) * mu]
                        rb_h = torch.empty(cnt * mu, dtype=torch.float64) if host_staging else None
                        plan.append((rank, sb, rb, rb_h))
                    plans[mu] = plan
                ops = []
                for rank, sb, rb, rb_h in plan:
                    if host_staging:
                        ops.append(dist.P2POp(dist.isend, sb.cpu(), rank))
                        ops.append(dist.P2POp(dist.irecv, rb_h, rank))
                    else:
                        ops.append(dist.P2POp(dist.isend, sb, rank))
                        ops.append(dist.P2POp(dist.irecv, rb, rank))
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
                if host_staging:
                    for rank, sb, rb, rb_h in plan:
                        rb.copy_(rb_h)
                torch.cuda.synchronize(device)
                return 0
            except Exception as e:  # never let an exception cross the C boundary
                print("halo transport failed:", e, flush=True)
                return -1

        def allreduce(ctx, buf, count):
            try:
                host = np.ctypeslib.as_array(buf, shape=(count,))
                if count > self._red.numel():
                    self._red = torch.zeros(count, dtype=torch.float64, device=self._red.device)
                t = self._red[:count]
                t.copy_(torch.from_numpy(host))
                dist.all_reduce(t)
                host[:] = t.cpu().numpy()
                return 0
            except Exception as e:
                print("all-reduce failed:", e, flush=True)
                return -1

        self._halo_cb = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int)(halo)
        self._red_cb = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.c_int)(allreduce)
        check(self._lib.HpddmHipSchwarzSetTransport(self._h, ctypes.cast(self._halo_cb, ctypes.c_void_p), ctypes.cast(self._red_cb, ctypes.c_void_p), None,
                                                    ctypes.c_void_p(self._send.data_ptr()), ctypes.c_void_p(self._recv.data_ptr()), mu_cap))

    # -- batched layout helpers --
    def pack(self, xs):
        """list of per-subdomain (n_s,) / (n_s, mu) arrays -> one flat array in the library's batched layout."""
        xs = [_as_f(x, np.complex128 if self.complex else np.float64) for x in xs]
        mu = 1 if xs[0].ndim == 1 else xs[0].shape[1]
        flat = np.concatenate([x.reshape(-1, order="F") for x in xs])
        return (flat.view(np.float64) if self.complex else flat), mu   # complex blocks travel as interleaved (re, im) doubles

    def unpack(self, flat, mu):
        if self.complex:
            flat = flat.view(np.complex128)
        out, off = [], 0
        for n in self.n:
            blk = flat[off:off + n * mu]
            out.append(blk.copy() if mu == 1 else blk.reshape(n, mu, order="F").copy(order="F"))
            off += n * mu
        return out

    def _op(self, fn, xs):
        flat, mu = self.pack(xs)
        out = np.empty_like(flat)
        check(fn(self._h, _dptr(flat), _dptr(out), mu))
        return self.unpack(out, mu)

    # -- the hot-path operations on host arrays --
    def exchange(self, xs):
        """schwarzExchange: returns sum_j R^T D x (the reference works in place)."""
        flat, mu = self.pack(xs)
        check(self._lib.HpddmHipSchwarzExchange(self._h, _dptr(flat), mu))
        return self.unpack(flat, mu)

    def gmv(self, xs):
        return self._op(self._lib.HpddmHipSchwarzGMV, xs)

    def apply(self, xs):
        return self._op(self._lib.HpddmHipSchwarzApply, xs)

    def deflation(self, xs):
        return self._op(self._lib.HpddmHipSchwarzDeflation, xs)

    def local_solve(self, xs):
        return self._op(self._lib.HpddmHipSchwarzLocalSolve, xs)

    def compute_residual(self, sol, f, norm="l2"):
        """schwarzComputeResidual: array of 2*mu values, [||f||, ||A x - f||] per right-hand side; norm: l2 | l1 | linfty."""
        fs, mu = self.pack(f)
        ss, _ = self.pack(sol)
        storage = np.zeros(2 * mu)
        check(self._lib.HpddmHipSchwarzComputeResidualNorm(self._h, _dptr(ss), _dptr(fs), _dptr(storage), mu, {"l2": 0, "l1": 1, "linfty": 2}[norm]))
        return storage

    def solve(self, f, sol=None, history=False):
        """solve(A, f, sol, comm) (interface/hpddm.py:267 -> HpddmSolve): returns (iterations, sol[, history])."""
        fs, mu = self.pack(f)
        xs = np.zeros_like(fs) if sol is None else self.pack(sol)[0]
        hist = np.zeros(4096)
        it = self._lib.HpddmHipSolve(self._h, _dptr(fs), _dptr(xs), mu, _dptr(hist), hist.size)
        if it < 0:
            raise HpddmHipError(self._lib.HpddmHipLastError().decode())
        out = self.unpack(xs, mu)
        return (it, out, hist[:it]) if history else (it, out)

    def set_custom_operator(self, mv, precond=None):
        """HpddmCustomOperatorSolve (interface/HPDDM.h:115): solve() then iterates on mv(in, out) as the operator and precond(in, out)
        as the preconditioner -- callables on host arrays of shape (n, mu), Fortran-ordered, `out` to be filled in place -- instead
        of the matrices and factors of this object.  None restores the Schwarz operator."""
        assert self.nsub == 1, "a custom operator is one block of rows per rank"

        def wrap(fn):
            if fn is None:
                return None

            def cb(_ctx, pin, pout, mu):
                try:
                    rows = self._custom_n
                    if self.complex:   # the library hands over (re, im) pairs: complex views of the same memory
                        xin = np.ctypeslib.as_array(pin, shape=(2 * mu * rows,)).view(np.complex128).reshape((rows, mu), order="F")
                        xout = np.ctypeslib.as_array(pout, shape=(2 * mu * rows,)).view(np.complex128).reshape((rows, mu), order="F")
                    else:
                        xin = np.ctypeslib.as_array(pin, shape=(mu * rows,)).reshape((rows, mu), order="F")
                        xout = np.ctypeslib.as_array(pout, shape=(mu * rows,)).reshape((rows, mu), order="F")
                    fn(xin, xout)
                    return 0
                except Exception:   # never unwind through the C frames
                    import traceback
                    traceback.print_exc()
                    return 1
            return ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.c_int)(cb)
        self._custom_n = int(self.n[0])
        self._custom_cbs = (wrap(mv), wrap(precond))   # kept alive with the operator
        as_ptr = lambda c: ctypes.cast(c, ctypes.c_void_p) if c is not None else None
        check(self._lib.HpddmHipSchwarzSetCustomOperator(self._h, as_ptr(self._custom_cbs[0]), as_ptr(self._custom_cbs[1]), None))

    # -- device-resident variants (raw HBM pointers) --
    def apply_host_flat(self, flat_in, flat_out, mu=1):
        """HpddmHipSchwarzApply on flat host arrays in the batched layout (the host-pointer boundary: both vectors cross PCIe)"""
        check(self._lib.HpddmHipSchwarzApply(self._h, _dptr(flat_in), _dptr(flat_out), mu))

    def apply_device(self, in_ptr, out_ptr, mu=1):
        check(self._lib.HpddmHipSchwarzApplyDevice(self._h, ctypes.c_void_p(in_ptr), ctypes.c_void_p(out_ptr), mu))

    def gmv_device(self, in_ptr, out_ptr, mu=1):
        check(self._lib.HpddmHipSchwarzGMVDevice(self._h, ctypes.c_void_p(in_ptr), ctypes.c_void_p(out_ptr), mu))

    def solve_device(self, b_ptr, x_ptr, mu=1):
        it = self._lib.HpddmHipSolveDevice(self._h, ctypes.c_void_p(b_ptr), ctypes.c_void_p(x_ptr), mu, None, 0)
        if it < 0:
            raise HpddmHipError(self._lib.HpddmHipLastError().decode())
        return it

    def synchronize(self):
        check(self._lib.HpddmHipSynchronize())

    def time(self, what, mu=1, warmup=2, reps=10):
        sec = ctypes.c_double()
        check(self._lib.HpddmHipSchwarzTime(self._h, what.encode(), mu, warmup, reps, ctypes.byref(sec)))
        return sec.value

    def rebuild_plan(self):
        """developer aid: build the SpTRSV level schedule again (its HPDDM_HIP_* knobs are read from the environment)"""
        check(self._lib.HpddmHipSchwarzRebuildPlan(self._h))

    def level_times(self, mu=1, reps=5):
        """developer aid: [(kind, level, microseconds, panel bytes)] of every launch of one batched SpTRSV; kind is one of
        'perm_in', 'gather', 'fwd', 'bwd', 'perm_out'"""
        out = np.zeros(3 * 512)
        n = self._lib.HpddmHipSchwarzLevelTimes(self._h, mu, reps, _dptr(out), out.size)
        if n < 0:
            raise HpddmHipError(self._lib.HpddmHipLastError().decode())
        kinds = ("perm_in", "gather", "fwd", "bwd", "perm_out")
        return [(kinds[int(out[3 * i]) // 1000], int(out[3 * i]) % 1000, out[3 * i + 1], out[3 * i + 2]) for i in range(n)]

    def stats(self):
        st = np.zeros(8)
        check(self._lib.HpddmHipSchwarzStats(self._h, _dptr(st)))
        keys = ("n", "nnz_L", "stored", "sptrsv_bytes_alg", "levels", "launches", "nnz_A", "coarse_dim")
        return dict(zip(keys, st))

    def subdomain(self, s):
        h = self._lib.HpddmHipSchwarzGetSubdomain(self._h, s)
        return Subdomain._borrow(h)

    def destroy(self):
        if self._h:
            self._lib.HpddmHipSchwarzDestroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def schwarz_from_subdomains(subs, first_global=0, nglobal=None, options="", multiplicity=True, partition=None):
    """examples/schwarz.cpp:90-97 in one call: create, set every subdomain, multiplicityScaling, initialize.
    partition = (rank, firsts) when the subdomains are sharded over several GPUs (then pass multiplicity=False and the
    final partition of unity in sd["d"])."""
    A = Schwarz(len(subs), first_global, nglobal)
    if options:
        A.option_parse(options)
    if partition is not None:
        A.set_partition(*partition)
    for s, sd in enumerate(subs):
        A.set_subdomain(s, sd["n"], sd["ia"], sd["ja"], sd["a"], sd["sym"], sd["neighbors"], sd["connectivity"], sd.get("numbering", "C"))
    d = [np.array(sd["d"], dtype=np.float64) for sd in subs]
    if multiplicity:
        d = A.multiplicity_scaling(d)
    A.initialize(d)
    return A, d
