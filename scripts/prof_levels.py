"""Per-launch durations of the last SpTRSV in a rocprofv3 rocpd database (development aid).
usage: prof_levels.py results.db <launches per solve> [level_stats.txt]
With the per-level panel sizes written by HPDDM_HIP_LEVEL_STATS=<file> the table also gives the read bandwidth of every level."""
import collections
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name,start,end,duration,grid_x,workgroup_x from kernels order by start").fetchall()
sp = [r for r in rows if 'sptrsv' in r[0]]
n = int(sys.argv[2])
last = sp[-n:]
rd = collections.defaultdict(float)
if len(sys.argv) > 3:
    for ln in open(sys.argv[3]):
        v = ln.split()
        rd[int(v[0])] += float(v[7]) * 8.0
nlev = max(rd) + 1 if rd else 0
t0 = last[0][1]
tot = 0
seen = {"fwd": 0, "bwd": 0}
for r in last:
    nm = r[0].split('::')[1].split('<')[0].replace('sptrsv_', '').replace('_kernel', '')
    extra = ""
    if rd and nm in ("fwd", "bwd"):
        # forward launches climb the levels, backward ones descend (levels without tiles have no launch: only valid when every level has one)
        lev = seen[nm] if nm == "fwd" else nlev - 1 - seen[nm]
        seen[nm] += 1
        extra = f"  level {lev:2d} {rd[lev] / 1e6:8.1f} MB {rd[lev] / r[3] :7.2f} GB/s"
    print(f"{nm:12s} start {(r[1]-t0)/1e3:8.1f}us dur {r[3]/1e3:8.1f}us wgs {r[4]//r[5]:7d}{extra}")
    tot += r[3]
print('sum', tot / 1e3, 'us; span', (last[-1][2] - t0) / 1e3, 'us')
