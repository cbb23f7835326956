"""MFMA utilisation per kernel from a scripts/pmc_summary.py table of a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
SQ_WAVE_CYCLES GRBM_GUI_ACTIVE` pass: SQ counters come as one row per shader engine (32 per dispatch on the 8 XCDs of an MI355X),
GRBM_GUI_ACTIVE as one per XCD (8 per dispatch).  utilisation = MFMA busy cycles summed over the chip / (1024 SIMDs x active cycles)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
by = {}
for r in rows:
    by.setdefault(r["kernel"], {})[r["counter"]] = (float(r["dispatches"]), float(r["sum"]))
print("kernel,kernel_dispatches,active_cycles_per_dispatch,mfma_busy_cycles_per_dispatch_whole_chip,mfma_busy_fraction_of_simd_cycles")
for k, c in by.items():
    if "GRBM_GUI_ACTIVE" not in c or "SQ_VALU_MFMA_BUSY_CYCLES" not in c:
        continue
    nd = c["GRBM_GUI_ACTIVE"][0] / 8.0
    act = c["GRBM_GUI_ACTIVE"][1] / c["GRBM_GUI_ACTIVE"][0]
    busy = c["SQ_VALU_MFMA_BUSY_CYCLES"][1] / nd
    print(f"\"{k}\",{nd:.0f},{act:.6g},{busy:.6g},{busy / 1024.0 / act:.4f}")
