#!/usr/bin/env python3
"""developer aid: where a narrow forward wave tile spends its time (HPDDM_HIP_DBG=32: clocks recorded by the tiles themselves).
usage: timeline.py [grid=128]"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["HPDDM_HIP_DBG"] = "32"
os.environ["HPDDM_HIP_STREAMS"] = "1"
from hpddm_amd import _lib, hpddm  # noqa: E402
from hpddm_amd.generate import generate3d  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
subs = generate3d(N, 8, overlap=1, sym=True, rhs="smooth")
A, d = hpddm.schwarz_from_subdomains(subs, options="-hpddm_operator_spd")
A.call_numfact()
A.time("solve", mu=1, warmup=1, reps=1)
A.rebuild_plan()           # resets the counter
A.time("solve", mu=1, warmup=0, reps=1)
lib = _lib.load()
cap = 1 << 20
buf = np.zeros(8 * cap, dtype=np.uint64)
n = lib.HpddmHipDebugTimeline(buf.ctypes.data_as(ctypes.c_void_p), cap)
t = buf[:8 * min(n, cap)].reshape(-1, 8).astype(np.float64)
print("tiles recorded", n)
dt = np.diff(t[:, :5], axis=1) * 0.01   # 100 MHz -> microseconds
names = ["entry->descriptor", "descriptor->rhs staged", "staged->panel streamed", "streamed->stored(acked)"]
for has_src in (0, 1):
    m = t[:, 7] == has_src
    if not m.any():
        continue
    kb = t[m, 5] * t[m, 6] * 8 / 1e3
    print(f"--- tiles {'with' if has_src else 'without'} children: {m.sum()}, panel KB median {np.median(kb):.1f}")
    for k, nm in enumerate(names):
        v = dt[m, k]
        print(f"   {nm:28s} median {np.median(v):6.2f} us   mean {v.mean():6.2f}   p90 {np.percentile(v, 90):6.2f}")
    tot = (t[m, 4] - t[m, 0]) * 0.01
    print(f"   {'total':28s} median {np.median(tot):6.2f} us   mean {tot.mean():6.2f}")

# per launch: the workgroups of a persistent launch start together, so the kernel-entry clocks cluster per launch
order = np.argsort(t[:, 0], kind="stable")
ts = t[order]
cuts = np.nonzero(np.diff(ts[:, 0]) > 200)[0] + 1
print("--- per launch (forward, wave tiles only): tiles, span [ticks], sum of tile durations / span = wavefronts busy on average")
for k, seg in enumerate(np.split(ts, cuts)[:12]):
    span = seg[:, 4].max() - seg[:, 0].min()
    busy = (seg[:, 4] - seg[:, 1]).sum() / span
    print(f"   launch {k:2d}: {len(seg):6d} tiles  span {span:9.0f} ticks  tile median {np.median(seg[:, 4] - seg[:, 1]):7.0f} ticks  busy wavefronts {busy:8.1f}  first tile starts {np.median(seg[:, 1] - seg[:, 0]):6.0f} ticks after entry (median)")
