"""Batched-SpTRSV timing out of a rocprofv3 --kernel-trace rocpd database when the subdomains are swept as several groups on
several streams (the launches of the groups overlap: the sum of the kernel durations is NOT the elapsed time).
For every batched solve -- delimited by its k_perm_in / k_perm_out launches -- prints the elapsed span (first start to last end),
the union of the busy intervals and the sum of the kernel durations; then their averages.
usage: prof_sweeps.py results.db [ngroups=4]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
ng = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows = c.execute("select name, start, end from kernels where name like '%sptrsv%' or name like '%k_perm_in%' or name like '%k_perm_out%' order by start").fetchall()
solves, cur, outs = [], [], 0
for name, s, e in rows:
    cur.append((s, e, name))
    if "k_perm_out" in name:
        outs += 1
        if outs == ng:
            solves.append(cur)
            cur, outs = [], 0
print("solve,launches,span_us,busy_union_us,sum_of_durations_us")
tot = [0.0, 0.0, 0.0]
for k, sv in enumerate(solves):
    span = (max(e for _, e, _ in sv) - min(s for s, _, _ in sv)) / 1e3
    ssum = sum(e - s for s, e, _ in sv) / 1e3
    union, hi = 0.0, None
    for s, e, _ in sorted(sv):
        if hi is None or s > hi:
            union += e - s
            hi = e
        elif e > hi:
            union += e - hi
            hi = e
    union /= 1e3
    if k >= len(solves) - 12:
        print(f"{k},{len(sv)},{span:.1f},{union:.1f},{ssum:.1f}")
    if k >= len(solves) - 20: # the timed region at the end of the run (earlier solves interleave with the eigenproblems of the set-up, whose own local solves carry the same kernel names)
        for i, v in enumerate((span, union, ssum)):
            tot[i] += v
n = max(1, min(20, len(solves)))
print(f"# {len(solves)} batched solves of {ng} groups; the last {n}: average span {tot[0] / n:.1f} us, busy union {tot[1] / n:.1f} us, sum of kernel durations {tot[2] / n:.1f} us")
