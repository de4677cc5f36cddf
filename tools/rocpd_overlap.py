#!/usr/bin/env python3
"""Do the kernels of different HIP streams / hardware queues overlap on the device?  From a rocprofv3 (rocpd sqlite) kernel
trace: per queue the number of dispatches, the sum of their durations and the span they cover; over all queues the union of
the busy intervals, and the overlap factor = sum of durations / union (1.0: strictly one kernel at a time).  Also an excerpt of
the timeline (queue, start, end of consecutive dispatches) so that the interleaving can be read.
Usage: tools/rocpd_overlap.py results.db [out.md] [name-filter-regex]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
qcol = next((c for c in ("queue_id", "queue", "stream_id", "stream") if c in cols), None)
flt = re.compile(sys.argv[3]) if len(sys.argv) > 3 else None
sel = f"select name, start, end, {qcol if qcol else '0'} from kernels order by start"
rows = [r for r in cur.execute(sel) if flt is None or flt.search(r[0])]
lines = [f"columns of `kernels`: {', '.join(cols)}", "", f"queue column: `{qcol}`; {len(rows)} dispatches" + (f" matching /{sys.argv[3]}/" if flt else ""), ""]
if rows:
    t0 = rows[0][1]
    per = {}
    for name, s, e, q in rows:
        p = per.setdefault(q, [0, 0, s, e])
        p[0] += 1
        p[1] += e - s
        p[2], p[3] = min(p[2], s), max(p[3], e)
    lines += ["| queue | dispatches | sum of durations ms | span ms | busy share of its span |", "|---|---|---|---|---|"]
    for q, (n, tot, s, e) in sorted(per.items(), key=lambda kv: str(kv[0])):
        lines.append(f"| {q} | {n} | {tot / 1e6:.3f} | {(e - s) / 1e6:.3f} | {tot / max(e - s, 1):.3f} |")
    # union of busy intervals over all queues
    iv = sorted((s, e) for _, s, e, _ in rows)
    union, cs, ce = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s <= ce:
            ce = max(ce, e)
        else:
            union += ce - cs
            cs, ce = s, e
    union += ce - cs
    total = sum(e - s for _, s, e, _ in rows)
    # time with >= 2 kernels in flight
    ev = sorted([(s, 1) for _, s, e, _ in rows] + [(e, -1) for _, s, e, _ in rows])
    depth, last, multi = 0, ev[0][0], 0
    for t, d in ev:
        if depth >= 2:
            multi += t - last
        depth += d
        last = t
    lines += ["", f"sum of kernel durations {total / 1e6:.3f} ms; union of busy time {union / 1e6:.3f} ms; overlap factor (sum / union) {total / max(union, 1):.3f}; "
              f"time with two or more kernels in flight {multi / 1e6:.3f} ms = {multi / max(union, 1):.3f} of the busy time; span {(iv[-1][1] - t0) / 1e6:.3f} ms", ""]
    mid = len(rows) // 2
    lines += ["timeline excerpt (mid-run, microseconds from the first dispatch of the excerpt):", "", "| queue | start | end | kernel |", "|---|---|---|---|"]
    base = rows[mid][1]
    for name, s, e, q in rows[mid:mid + 48]:
        lines.append(f"| {q} | {(s - base) / 1e3:.2f} | {(e - base) / 1e3:.2f} | `{re.sub(r'zg::|.anonymous namespace.::', '', name)[:48]}` |")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
