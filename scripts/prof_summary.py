"""Text summary (per-kernel calls / total / average / share) of a rocprofv3 `--kernel-trace --stats` rocpd database."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"# rocprofv3 --kernel-trace --stats summary of {sys.argv[1]}")
print(f"# command: {' '.join(sys.argv[2:])}")
print("kernel,calls,total_ms,avg_us,min_us,max_us,percent")
for name, n, t, a, mn, mx in rows:
    print(f"\"{name}\",{n},{t/1e6:.3f},{a/1e3:.3f},{mn/1e3:.3f},{mx/1e3:.3f},{100*t/tot:.2f}")
