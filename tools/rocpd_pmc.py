#!/usr/bin/env python3
"""Per-kernel average of a PMC counter from a rocprofv3 rocpd sqlite db.  Usage: rocpd_pmc.py db [out.md]"""
import re, sqlite3, sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
rows = list(cur.execute("select * from counters_collection"))
ix = {c: i for i, c in enumerate(cols)}
name_col = "kernel_name" if "kernel_name" in ix else [c for c in cols if "kernel" in c and "name" in c][0]
cnt_col = "counter_name" if "counter_name" in ix else [c for c in cols if "counter" in c and "name" in c][0]
val_col = "value" if "value" in ix else [c for c in cols if "value" in c][0]
agg = {}
for r in rows:
    k = (re.sub(r"zg::\(anonymous namespace\)::", "", r[ix[name_col]])[:100], r[ix[cnt_col]])
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += float(r[ix[val_col]])
lines = ["| kernel | counter | dispatches | avg per dispatch | total |", "|---|---|---|---|---|"]
for (k, c), (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lines.append(f"| `{k}` | {c} | {n} | {tot / n:.1f} | {tot:.0f} |")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
