"""ctypes loader of libhpddm_hip.so (the C ABI declared in include/hpddm_hip.h).

The library is the product: there is no Python/CPU fallback.  Loading fails loudly if the shared object has not been
built (`python -c "import __graft_entry__ as g; g.build()"` or `make -C hpddm_amd/csrc`).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhpddm_hip.so")

c_double_p = ctypes.POINTER(ctypes.c_double)
c_int_p = ctypes.POINTER(ctypes.c_int)
c_ll_p = ctypes.POINTER(ctypes.c_longlong)
c_void_pp = ctypes.POINTER(ctypes.c_void_p)

_lib = None


class HpddmHipError(RuntimeError):
    pass


def _declare(lib):
    P, I, D, C = ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_char_p
    US, LL = ctypes.c_ushort, ctypes.c_longlong
    sig = {
        "HpddmHipLastError": (C, []),
        "HpddmHipDeviceCount": (I, []),
        "HpddmHipSetDevice": (I, [I]),
        "HpddmHipSynchronize": (I, []),
        "HpddmHipSubdomainNumfact": (I, [c_void_pp, I, P, P, P, I, ctypes.c_char, I]),
        "HpddmHipSubdomainNumfactZ": (I, [c_void_pp, I, P, P, P, I, ctypes.c_char, I]),
        "HpddmHipSubdomainSolve": (I, [P, P, P, US]),
        "HpddmHipSubdomainSolveZ": (I, [P, P, P, US]),
        "HpddmHipSubdomainSolveDevice": (I, [P, P, P, US]),
        "HpddmHipSubdomainDestroy": (None, [P]),
        "HpddmHipSubdomainSetOption": (I, [c_void_pp, C, D]),
        "HpddmHipSubdomainInertia": (I, [P]),
        "HpddmHipSubdomainRefineSteps": (I, [P]),
        "HpddmHipSubdomainInfo": (I, [P, P, P]),
        "HpddmHipSubdomainExport": (LL, [P, C, P, LL]),
        "HpddmHipSubdomainExportView": (P, [P, C, c_ll_p]),
        "HpddmHipSubdomainTimeSolve": (I, [P, I, I, I, P]),
        "HpddmHipSchwarzCreate": (P, [I, I, I]),
        "HpddmHipSchwarzDestroy": (None, [P]),
        "HpddmHipSchwarzSetSubdomain": (I, [P, I, I, P, P, P, I, ctypes.c_char, I, P, P, P]),
        "HpddmHipSchwarzMultiplicityScaling": (I, [P, P]),
        "HpddmHipSchwarzInitialize": (I, [P, I, P]),
        "HpddmHipSchwarzSetVectors": (I, [P, I, I, P]),
        "HpddmHipSchwarzSetSubdomainZ": (I, [P, I, I, P, P, P, I, ctypes.c_char, I, P, P, P]),
        "HpddmHipSchwarzSetVectorsZ": (I, [P, I, I, P]),
        "HpddmHipSchwarzIsComplex": (I, [P]),
        "HpddmHipDenseEig": (I, [I, P, P, P, P]),
        "HpddmHipDenseEigZ": (I, [I, P, P, P]),
        "HpddmHipHostSelfTest": (I, []),
        "HpddmHipSchwarzDestroyRecycling": (I, [P]),
        "HpddmHipSchwarzSolveGEVP": (I, [P, I, I, P, P, P, I, ctypes.c_char]),
        "HpddmHipSchwarzSetOptimizedMatrix": (I, [P, I, I, P, P, P, I, ctypes.c_char]),
        "HpddmHipSchwarzSetOptimizedMatrixZ": (I, [P, I, I, P, P, P, I, ctypes.c_char]),
        "HpddmHipSchwarzGetEigenvalues": (I, [P, I, P, I]),
        "HpddmHipSchwarzGetEigenvaluesZ": (I, [P, I, P, I]),
        "HpddmHipSchwarzGetVectors": (I, [P, I, P, ctypes.c_longlong]),
        "HpddmHipSchwarzSolveGEVPWith": (I, [P, I, I, P, P, P, I, ctypes.c_char, P, P, P, I]),
        "HpddmHipSchwarzBuildCoarseOperator": (I, [P]),
        "HpddmHipSchwarzCallNumfact": (I, [P]),
        "HpddmHipSchwarzSetOption": (I, [P, C, D]),
        "HpddmHipSchwarzGetOption": (D, [P, C]),
        "HpddmHipSchwarzOptionParse": (I, [P, C]),
        "HpddmHipSchwarzGetDof": (LL, [P, I]),
        "HpddmHipSchwarzExchange": (I, [P, P, US]),
        "HpddmHipSchwarzGMV": (I, [P, P, P, US]),
        "HpddmHipSchwarzApply": (I, [P, P, P, US]),
        "HpddmHipSchwarzDeflation": (I, [P, P, P, US]),
        "HpddmHipSchwarzLocalSolve": (I, [P, P, P, US]),
        "HpddmHipSchwarzComputeResidual": (I, [P, P, P, P, US]),
        "HpddmHipSchwarzComputeResidualNorm": (I, [P, P, P, P, US, I]),
        "HpddmHipSolve": (I, [P, P, P, I, P, I]),
        "HpddmHipSchwarzSetCustomOperator": (I, [P, P, P, P]),
        "HpddmHipSchwarzSetPartition": (I, [P, I, I, P]),
        "HpddmHipSchwarzHaloPeers": (I, [P, I, P, P, P]),
        "HpddmHipSchwarzSetTransport": (I, [P, P, P, P, P, P, I]),
        "HpddmHipSchwarzHaloExport": (LL, [P, C, P, LL]),
        "HpddmHipRcclGetUniqueId": (I, [P]),
        "HpddmHipSchwarzInitRccl": (I, [P, P, I]),
        "HpddmHipRcclSelfTest": (I, []),
        "HpddmHipRcclHaloProbe": (I, [P, P, P, P, I, P, P, LL]),
        "HpddmHipSchwarzApplyDevice": (I, [P, P, P, US]),
        "HpddmHipSchwarzGMVDevice": (I, [P, P, P, US]),
        "HpddmHipSolveDevice": (I, [P, P, P, I, P, I]),
        "HpddmHipSchwarzTime": (I, [P, C, I, I, I, P]),
        "HpddmHipSchwarzStats": (I, [P, P]),
        "HpddmHipSchwarzRebuildPlan": (I, [P]),
        "HpddmHipSchwarzLevelTimes": (I, [P, I, I, P, I]),
        "HpddmHipSchwarzGetSubdomain": (P, [P, I]),
        "HpddmHipPanelCreate": (P, [I, I, P, P]),
        "HpddmHipPanelCreateZ": (P, [I, I, P, P]),
        "HpddmHipPanelZtD": (I, [P, P, P, US]),
        "HpddmHipPanelZ": (I, [P, P, P, US]),
        "HpddmHipPanelDestroy": (None, [P]),
    }
    missing = []
    for name, (res, args) in sig.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    return sig, missing


#: every symbol include/hpddm_hip.h declares (tests check the library exports all of them)
DECLARED_SYMBOLS = None


def load():
    global _lib, DECLARED_SYMBOLS
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HpddmHipError(
                f"{LIB_PATH} is missing: the HIP library must be built first (python -c 'import __graft_entry__ as g; "
                "g.build()'). There is no CPU fallback.")
        lib = ctypes.CDLL(LIB_PATH)
        sig, missing = _declare(lib)
        DECLARED_SYMBOLS = sorted(sig)
        if missing:
            raise HpddmHipError(f"{LIB_PATH} does not export {missing}; rebuild it")
        _lib = lib
    return _lib


def check(code):
    if code is None:
        return
    if code < 0:
        raise HpddmHipError(load().HpddmHipLastError().decode())
    return code
