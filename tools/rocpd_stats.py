#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2 rocpd sqlite) kernel trace: per-kernel count / avg / min / max / total,
plus the inter-kernel gaps on the device timeline.  Usage: tools/rocpd_stats.py results.db [out.md]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"zg::\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(zg::[A-Za-z]+Args\)", "", name)
    return name[:110]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = list(cur.execute("select name, start, end from kernels order by start"))
    stats = {}
    for name, s, e in rows:
        st = stats.setdefault(name, [0, 0, 1 << 62, 0])
        d = e - s
        st[0] += 1
        st[1] += d
        st[2] = min(st[2], d)
        st[3] = max(st[3], d)
    total = sum(v[1] for v in stats.values())
    gaps = [rows[i + 1][1] - rows[i][2] for i in range(len(rows) - 1)]
    gaps_small = [g for g in gaps if 0 <= g < 50_000]
    lines = ["| kernel | calls | avg us | min us | max us | total ms | % |", "|---|---|---|---|---|---|---|"]
    for name, (n, tot, mn, mx) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"| `{short(name)}` | {n} | {tot / n / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {tot / 1e6:.2f} | {100 * tot / total:.1f} |")
    lines.append("")
    lines.append(f"kernel time total {total / 1e6:.2f} ms over {len(rows)} dispatches; "
                 f"inter-kernel gaps < 50 us: n={len(gaps_small)}, mean {sum(gaps_small) / max(len(gaps_small), 1) / 1e3:.2f} us, "
                 f"sum {sum(gaps_small) / 1e6:.2f} ms")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
