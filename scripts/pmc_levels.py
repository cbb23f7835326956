"""Per-launch value of one rocprofv3 --pmc counter over the last batched SpTRSV of a run (development aid / profiles).
usage: pmc_levels.py results.db <launches per solve> [level_stats.txt]"""
import collections
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select dispatch_id, name, counter_name, sum(counter_value), min(start) from pmc_events where name like '%sptrsv%' group by dispatch_id, name, counter_name order by dispatch_id").fetchall()
n = int(sys.argv[2])
last = rows[-n:]
stored = collections.defaultdict(float)
read = collections.defaultdict(float)
if len(sys.argv) > 3:
    for ln in open(sys.argv[3]):
        v = ln.split()
        stored[int(v[0])] += float(v[6]) * 8.0
        read[int(v[0])] += float(v[7]) * 8.0
nlev = max(read) + 1 if read else 0
seen = {"fwd": 0, "bwd": 0}
tot = 0.0
print("kind,level,counter,value_KB,panel_entries_MB,stored_padded_MB,value_over_entries")
for did, name, cn, val, _ in last:
    nm = name.split('::')[1].split('<')[0].replace('sptrsv_', '').replace('_kernel', '')
    lev = ""
    ratio = ""
    if read and nm in ("fwd", "bwd"):
        lev = seen[nm] if nm == "fwd" else nlev - 1 - seen[nm]
        seen[nm] += 1
        ratio = f"{val * 1024.0 / read[lev]:.3f}"
        print(f"{nm},{lev},{cn},{val:.0f},{read[lev] / 1e6:.1f},{stored[lev] / 1e6:.1f},{ratio}")
    else:
        print(f"{nm},,{cn},{val:.0f},,,")
    tot += val
print(f"# total over the sweep pair: {tot:.0f} KB")
