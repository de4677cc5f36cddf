#!/usr/bin/env python3
"""<dir>/<tag>_pmc_{FETCH,WRITE}_SIZE.md + <tag>_kernel_stats.md -> <dir>/traffic.json: HBM traffic per launch of every
kernel class of the 124M single-prompt decode step, corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE is in KB
and reports half of a wide coalesced read on gfx950: doubled; WRITE_SIZE as reported).  bench.py quotes the entry of
its dominant class with this provenance.

usage: python tools/make_traffic_json.py <dir with the md files> <tag>"""
import json
import sys

CLASSES = {  # bench.py kernel class -> kernel symbol (substring of the rocprof name) at 124M, one prompt, bf16 weights
    "ln_1 + c_attn + KV append": "gemv_lnk_kernel<unsigned short, 16, 2, 4>",
    "attention (split-KV decode)": "attn_decode_kernel<float>",
    "head merge + attn c_proj + residual": "gemv_ksplit_kernel<unsigned short, 16, 2, 2>",
    "ln_2 + c_fc + GELU": "gemv_lnk_kernel<unsigned short, 16, 2, 4>",
    "mlp c_proj + residual": "gemv_ksplit_kernel<unsigned short, 32, 3, 2>",
    "ln_f + lm_head + argmax": "gemv_kernel<unsigned short, 1, 16, 6, true>",
}
NOTE = {"gemv_lnk_kernel<unsigned short, 16, 2, 4>": "kernel shared by c_attn (3.54 MB) and c_fc (4.72 MB): average of both"}


def row(path, sym, col):
    for line in open(path):
        if sym in line:
            cells = [c.strip() for c in line.strip().strip("|").split("|")]
            return float(cells[col])
    raise SystemExit(f"{sym} not found in {path}")


def main():
    d, tag = sys.argv[1], sys.argv[2]
    out = {}
    for cls, sym in CLASSES.items():
        fetch = row(f"{d}/{tag}_pmc_FETCH_SIZE.md", sym, 3)
        write = row(f"{d}/{tag}_pmc_WRITE_SIZE.md", sym, 3)
        us = row(f"{d}/{tag}_kernel_stats.md", sym, 2)
        out[cls] = {
            "kernel": sym, "FETCH_SIZE_avg_KB": fetch, "WRITE_SIZE_avg_KB": write,
            "bytes_per_launch": int(round((2 * fetch + write) * 1024)), "rocprof_avg_us_in_situ": us,
            "source": f"profiles/{tag}_pmc_FETCH_SIZE.md + profiles/{tag}_pmc_WRITE_SIZE.md (rocprofv3 --pmc, separate passes, full "
                      f"1024-position context, eager launches; FETCH_SIZE doubled per MI355X_MICROARCH.md, KB units)"
                      + (f"; {NOTE[sym]}" if sym in NOTE else ""),
        }
    json.dump({"124M/1/bf16": out}, open(f"{d}/traffic.json", "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
