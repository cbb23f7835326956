"""Batched-SpTRSV timing out of a rocprofv3 --kernel-trace rocpd database when the subdomains are swept as several groups on
several streams (the launches of the groups overlap: the sum of the kernel durations is NOT the elapsed time).
For every batched solve -- delimited by its k_perm_in / k_perm_out launches -- prints the elapsed span (first start to last end),
the union of the busy intervals and the sum of the kernel durations; then their averages.
usage: prof_sweeps.py results.db [ngroups=4]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
ng = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
sid = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
rows = c.execute(f"select name, start, end, {sid or 0} from kernels where name like '%sptrsv%' or name like '%k_root_sym%' or name like '%k_root_reduce%' or name like '%k_perm_in%' or name like '%k_perm_out%' order by start").fetchall()
solves = []
if sid:
    # every group sweeps on its own stream: the k-th k_perm_in of a stream opens the k-th solve of that stream, and the k-th solves of the
    # ng streams that make one batched solve are the ones whose k_perm_in launches are closest in time (round 5: cutting the time-ordered
    # list at every ng-th k_perm_out mis-assigned kernels at the boundaries once the groups' streams were picked by timing -- 229 .. 242
    # launches per solve where the plan has 236)
    per = {}
    for name, s, e, q in rows:
        lst = per.setdefault(q, [])
        if "k_perm_in" in name or not lst:
            lst.append([])
        lst[-1].append((s, e, name))
    streams = [v for v in per.values() if len(v) > 1]
    streams.sort(key=len, reverse=True)
    streams = streams[:ng]
    k = min(len(v) for v in streams) if streams else 0
    for j in range(1, k + 1):   # align from the END of the run (the timed region): the set-up makes solves of its own on the library stream
        solves.append([t for v in streams for t in v[-j]])
    solves.reverse()
else:
    cur, outs = [], 0
    for name, s, e, _ in rows:
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
