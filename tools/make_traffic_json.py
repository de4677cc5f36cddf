#!/usr/bin/env python3
"""<dir>/<tag>_<cfg>_pmc_{FETCH,WRITE}_SIZE.md + <tag>_<cfg>_kernel_stats.md -> <dir>/traffic.json: HBM traffic per launch of every
kernel class of the decode step, per benchmark configuration, corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE is in KB
and reports half of a wide coalesced read on gfx950: doubled; WRITE_SIZE as reported).  bench.py quotes the entry of its dominant
class with this provenance (`roofline.traffic`, `traffic_source`).

usage: python tools/make_traffic_json.py <dir with the md files> <tag>
Configurations: 124m (one prompt), 124m_8prompts, xl — keys "124M/1/bf16", "124M/8/bf16", "xl/1/bf16" of traffic.json.
A kernel symbol that serves several classes (the batched Linears share instantiations) gets the average of its launches, noted."""
import json
import re
import sys

# bench.py kernel class -> regex of the rocprof kernel name, per configuration (first table row that matches)
CLASSES = {
    "124M/1/bf16": ("124m", {
        "ln_1 + c_attn + KV append": r"gemv_lnk_kernel<unsigned short, 16, 2, 4>",
        "attention (split-KV decode)": r"attn_decode_kernel<float>",
        "head merge + attn c_proj + residual": r"gemv_ksplit_kernel<unsigned short, 16, 2, 2>",
        "ln_2 + c_fc + GELU": r"gemv_lnk_kernel<unsigned short, 16, 2, 4>",
        "mlp c_proj + residual": r"gemv_ksplit_kernel<unsigned short, 32, 3, 2>",
        "ln_f + lm_head + argmax": r"gemv_kernel<unsigned short, 1, 16, 6, true>",
    }),
    "124M/8/bf16": ("124m_8prompts", {
        "ln_1 + c_attn + KV append": r"gemv_pl4_kernel<3, 1>",
        "attention (split-KV decode)": r"attn_decode_kernel<float>",
        "head merge + attn c_proj + residual": r"gemv_pl4_kernel<3, 1>",
        "ln_2 + c_fc + GELU": r"gemv_pl4_kernel<3, 1>",
        "mlp c_proj + residual": r"gemv_pl4_kernel<3, 4>",
        "ln_f + lm_head + argmax": r"lm_head_wpt_kernel",
    }),
    "xl/1/bf16": ("xl", {
        "ln_1 + c_attn + KV append": r"gemv_lnk_kernel<unsigned short, 32, 2, 4>",
        "attention (split-KV decode)": r"attn_decode_kernel<float>",
        "head merge + attn c_proj + residual": r"gemv_kernel<unsigned short, 1, 32, 8, false>",
        "ln_2 + c_fc + GELU": r"gemv_lnk_kernel<unsigned short, 32, 2, 4>",
        "mlp c_proj + residual": r"gemv_ksplit_kernel<unsigned short, 32, 7, 2>",
        "ln_f + lm_head + argmax": r"gemv_kernel<unsigned short, 1, 32, 8, true>",
    }),
}


def row(path, pattern, col):
    rx = re.compile(pattern)
    for line in open(path):
        if rx.search(line):
            cells = [c.strip() for c in line.strip().strip("|").split("|")]
            return float(cells[col]), cells[0].strip("`")[:90]
    return None, None


def main():
    d, tag = sys.argv[1], sys.argv[2]
    result = {}
    for key, (cfg, classes) in CLASSES.items():
        out = {}
        try:
            users = {}
            for cls, pat in classes.items():
                users.setdefault(pat, []).append(cls)
            for cls, pat in classes.items():
                fetch, sym = row(f"{d}/{tag}_{cfg}_pmc_FETCH_SIZE.md", pat, 3)
                write, _ = row(f"{d}/{tag}_{cfg}_pmc_WRITE_SIZE.md", pat, 3)
                us, _ = row(f"{d}/{tag}_{cfg}_kernel_stats.md", pat, 2)
                if fetch is None or write is None:
                    continue
                note = f"; kernel shared by {', '.join(users[pat])}: average over its launches" if len(users[pat]) > 1 else ""
                out[cls] = {
                    "kernel": sym, "FETCH_SIZE_avg_KB": fetch, "WRITE_SIZE_avg_KB": write,
                    "bytes_per_launch": int(round((2 * fetch + write) * 1024)), "rocprof_avg_us_in_situ": us,
                    "source": f"profiles/{tag}_{cfg}_pmc_FETCH_SIZE.md + profiles/{tag}_{cfg}_pmc_WRITE_SIZE.md (rocprofv3 --pmc, separate passes, "
                              f"full-context generation, eager launches; FETCH_SIZE doubled per MI355X_MICROARCH.md, KB units)" + note,
                }
        except FileNotFoundError as e:
            print(f"{key}: {e}", file=sys.stderr)
        if out:
            result[key] = out
    json.dump(result, open(f"{d}/traffic.json", "w"), indent=1)
    print(json.dumps(result))


if __name__ == "__main__":
    main()
