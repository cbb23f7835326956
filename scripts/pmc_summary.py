"""Sum a rocprofv3 --pmc counter per kernel name from a rocpd database (development aid / profiles)."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, counter_name, count(*), sum(counter_value), sum(duration) from pmc_events group by name, counter_name order by sum(counter_value) desc").fetchall()
print("kernel,counter,dispatches,sum,sum_per_dispatch,total_ms")
for name, cn, n, v, dur in rows:
    short = name.split("(")[0][-60:]
    print(f"\"{short}\",{cn},{n},{v:.6g},{v/n:.6g},{dur/1e6:.3f}")
