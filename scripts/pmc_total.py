"""Sum of one rocprofv3 --pmc counter over the kernels of the last batched SpTRSV of a run (all groups / streams).
usage: pmc_total.py results.db [ngroups=4]  -> prints  <counter> <sum over the last solve> <dispatches>"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
ng = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows = c.execute("select dispatch_id, name, counter_name, sum(counter_value), min(start) from pmc_events where name like '%sptrsv%' or name like '%k_root_sym%' or name like '%k_root_reduce%' or name like '%k_perm_in%' or name like '%k_perm_out%' "
                 "group by dispatch_id, name, counter_name order by min(start)").fetchall()
# walk backwards to the ng-th k_perm_in from the end
ins, first = 0, 0
for i in range(len(rows) - 1, -1, -1):
    if "k_perm_in" in rows[i][1]:
        ins += 1
        if ins == ng:
            first = i
            break
last = rows[first:]
per = {}
for _, name, cn, val, _ in last:
    short = name.split("::")[-1].split("<")[0].split("(")[0]
    per[short] = per.get(short, 0.0) + val
print(f"{last[0][2]} {sum(r[3] for r in last):.0f} {len(last)}")
for k, v in sorted(per.items(), key=lambda kv: -kv[1]):
    print(f"#  {k}: {v:.0f}")
