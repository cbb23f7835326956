#!/usr/bin/env python3
"""SQ counters of the 16-column engine (two rocprofv3 --pmc passes, scripts/r06_profiles.sh sq16) -> per kernel: dispatches, wavefronts per
dispatch, share of the wave cycles spent waiting, wavefronts alive per SIMD (counters of ONE shader engine = 1/32 of the chip, per dispatch;
SQ_WAVE_CYCLES / SQ_WAIT_* count in units of 4 cycles, SQ_BUSY_CYCLES in cycles: profiles/r05_engine16_sq_counters.txt)."""
import csv
import sys

vals = {}
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        vals.setdefault(r["kernel"], {})[r["counter"]] = (int(r["dispatches"]), float(r["sum"]), float(r["total_ms"]))
print("# 16-column engine, Helmholtz share of configs[4], 8 complex right-hand sides, one group on one stream (scripts/sweep_plan.py --helmholtz 64,64,128 --mu 8)")
for k, v in sorted(vals.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", (0, 0, 0))[1]):
    g = lambda c: v.get(c, (0, 0.0, 0.0))
    n, wc, ms = g("SQ_WAVE_CYCLES")
    if not n:
        continue
    waves, busy, wait_any = g("SQ_WAVES")[1], g("SQ_BUSY_CYCLES")[1], g("SQ_WAIT_ANY")[1]
    wi, ai = g("SQ_WAIT_INST_ANY")[1], g("SQ_ACTIVE_INST_ANY")[1]
    print(f"{k}: {n} dispatches, {ms / n * 1e3:.1f} us each under the counters; wavefronts per dispatch and shader engine {waves / n:.0f}")
    print(f"    share of the wave cycles: waiting for anything {wait_any / wc:.2f}, waiting for issue {wi / wc:.2f}, executing {ai / wc:.2f}; "
          f"wavefronts alive per SIMD {4.0 * wc / busy / 32.0 * 4.0 / 4.0:.1f}; wave cycles per wavefront {4.0 * wc / max(waves, 1):.0f}; "
          f"vector loads per wavefront {g('SQ_INSTS_VMEM_RD')[1] / max(waves, 1):.0f}, MFMAs {g('SQ_INSTS_MFMA')[1] / max(waves, 1):.0f}")
