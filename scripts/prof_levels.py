"""Per-launch durations of the last SpTRSV in a rocprofv3 rocpd database (development aid)."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name,start,end,duration,grid_x,workgroup_x from kernels order by start").fetchall()
sp = [r for r in rows if 'sptrsv' in r[0]]
n = int(sys.argv[2])
last = sp[-n:]
t0 = last[0][1]
tot = 0
for r in last:
    nm = r[0].split('::')[1].split('<')[0].replace('sptrsv_', '').replace('_kernel', '')
    print(f"{nm:9s} start {(r[1]-t0)/1e3:8.1f}us dur {r[3]/1e3:8.1f}us wgs {r[4]//r[5]:7d}")
    tot += r[3]
print('sum', tot / 1e3, 'us; span', (last[-1][2] - t0) / 1e3, 'us')
