"""hpddm_amd: MI355X-native Restricted Additive Schwarz apply behind HPDDM's Solver / Schwarz / C-ABI surface.

The compute lives in ``libhpddm_hip.so`` (hand-written HIP for gfx950, C ABI in ``include/hpddm_hip.h``); this
package is the thin host side: the ctypes binding mirroring the reference's ``interface/hpddm.py`` and the problem
generators mirroring ``examples/generate.py``.
"""
from . import generate  # noqa: F401
from ._lib import HpddmHipError, LIB_PATH  # noqa: F401
