#!/usr/bin/env python3
"""profiles/<tag>_pmc_{FETCH,WRITE}_SIZE.md + <tag>_kernel_stats.md -> profiles/lm_head_traffic.json
(the HBM traffic per launch of bench.py's roofline kernel, corrected as MI355X_MICROARCH.md prescribes:
FETCH_SIZE is reported in KB and must be doubled on gfx950, WRITE_SIZE is taken as reported).

usage: python tools/make_traffic_json.py <dir with the md files> <tag>"""
import json
import re
import sys

KERNEL = "gemv_kernel<unsigned short, 1, 16, 6, true>"


def row(path, col):
    for line in open(path):
        if KERNEL in line:
            cells = [c.strip() for c in line.strip().strip("|").split("|")]
            return float(cells[col])
    raise SystemExit(f"{KERNEL} not found in {path}")


def main():
    d, tag = sys.argv[1], sys.argv[2]
    fetch = row(f"{d}/{tag}_pmc_FETCH_SIZE.md", 3)
    write = row(f"{d}/{tag}_pmc_WRITE_SIZE.md", 3)
    us = row(f"{d}/{tag}_kernel_stats.md", 2)
    out = {
        "kernel": "gemv_kernel<bf16,M=1,LPR16,CPL6,ARGMAX> (ln_f + lm_head + argmax)",
        "round_profile": tag,
        "FETCH_SIZE_avg_KB": fetch,
        "WRITE_SIZE_avg_KB": write,
        "traffic_bytes_per_launch": int(round((2 * fetch + write) * 1024)),
        "correction": "FETCH_SIZE doubled (gfx950 reports half of a wide coalesced read, MI355X_MICROARCH.md HBM "
                      "section); WRITE_SIZE as reported; counters in KB",
        "rocprof_avg_us": us,
        "source": f"profiles/{tag}_pmc_FETCH_SIZE.md, profiles/{tag}_pmc_WRITE_SIZE.md, profiles/{tag}_kernel_stats.md",
    }
    json.dump(out, open(f"{d}/lm_head_traffic.json", "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
