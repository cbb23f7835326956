"""Raw-ctypes GPU check of the local solver (HIP SpTRSV) against scipy on 3-D Poisson; development aid."""
import ctypes, sys, time, os
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'hpddm_amd', 'libhpddm_hip.so'))
lib.HpddmHipLastError.restype = ctypes.c_char_p
lib.HpddmHipSubdomainSetOption.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_char_p, ctypes.c_double]
lib.HpddmHipSubdomainNumfact.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_char, ctypes.c_int]
lib.HpddmHipSubdomainSolve.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ushort]
lib.HpddmHipSubdomainInfo.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
lib.HpddmHipSubdomainTimeSolve.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]

def poisson3d(N):
    I = sp.identity(N); T = sp.diags([-1, 2, -1], [-1, 0, 1], shape=(N, N))
    return (sp.kron(sp.kron(T, I), I) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(I, I), T)).tocsr()

print('devices', lib.HpddmHipDeviceCount())
ok = True
cases = [(8, 'chol', [1, 2, 3, 4, 5, 8, 11]), (12, 'ldlt', [1, 4]), (11, 'lu', [1, 2]), (24, 'chol', [1, 8]), (40, 'chol', [1])] + ([(65, 'chol', [1, 4])] if len(sys.argv) > 1 else [])
if len(sys.argv) > 2:
    cases = [(int(v), 'chol', [1]) for v in sys.argv[2:]]
for N, mode, mus in cases:
    A = poisson3d(N)
    if mode == 'lu':
        rng = np.random.default_rng(0)
        A = (A + sp.diags(rng.random(A.shape[0])) + sp.triu(A, 1) * 0.3).tocsr(); Ain, sym, spd = A, 0, 0
    elif mode == 'ldlt':
        A = (A - 1.7 * sp.identity(A.shape[0])).tocsr(); Ain, sym, spd = sp.tril(A).tocsr(), 1, 0
    else:
        Ain, sym, spd = sp.tril(A).tocsr(), 1, 1
    Ain.sort_indices(); n = A.shape[0]
    S = ctypes.c_void_p()
    ia = Ain.indptr.astype(np.int32); ja = Ain.indices.astype(np.int32); a = Ain.data.astype(np.float64)
    t0 = time.time()
    rc = lib.HpddmHipSubdomainNumfact(ctypes.byref(S), n, ia.ctypes.data, ja.ctypes.data, a.ctypes.data, sym, b'C', spd)
    assert rc == 0, lib.HpddmHipLastError()
    info = np.zeros(12, dtype=np.int64); times = np.zeros(4)
    lib.HpddmHipSubdomainInfo(S, info.ctypes.data, times.ctypes.data)
    print(f'N={N} {mode}: n={n} nnzL={info[3]} levels={info[2]} launches={info[8]} numfact {time.time()-t0:.2f}s times={times}')
    for mu in mus:
        b = np.random.default_rng(mu).random((mu, n))
        x = np.zeros_like(b)
        rc = lib.HpddmHipSubdomainSolve(S, b.ctypes.data, x.ctypes.data, mu)
        assert rc == 0, lib.HpddmHipLastError()
        res = max(np.linalg.norm(A @ x[k] - b[k]) / np.linalg.norm(b[k]) for k in range(mu))
        flag = 'OK' if res < 1e-10 else 'FAIL'
        ok &= res < 1e-10
        sec = ctypes.c_double()
        lib.HpddmHipSubdomainTimeSolve(S, mu, 2, 10, ctypes.byref(sec))
        bytes_alg = 2 * info[3] * 8 + 4 * n * mu * 8
        print(f'   mu={mu}: rel residual {res:.2e} {flag}; solve {sec.value*1e3:.3f} ms -> {bytes_alg/sec.value/1e9:.1f} GB/s algorithmic')
print('ALL OK' if ok else 'SOME FAILED')
