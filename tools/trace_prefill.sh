#!/bin/bash
# Usage (GPU box): tools/trace_prefill.sh <tag> <batch> [env assignments...] -> gpurun_out/<tag>_prefill_<batch>x1023_kernel_stats.md
# Kernel trace of the whole-prompt pass (tools/bench_prefill.py) through rocprofv3.
tag=$1; batch=$2; shift 2
out=gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
timeout 600 rocprofv3 --kernel-trace --stats -d $out/tr_$tag -o t -- python3 tools/bench_prefill.py --batch $batch --lengths 1023 --reps 10 > $out/${tag}_under_rocprof.json 2> $out/${tag}_under_rocprof.err
python tools/rocpd_stats.py $(find $out/tr_$tag -name "*.db" | head -1) $out/${tag}_prefill_${batch}x1023_kernel_stats.md > /dev/null
rm -rf $out/tr_$tag
grep -E "gemm|attn_prefill|reduce|ln_split" $out/${tag}_prefill_${batch}x1023_kernel_stats.md | cut -c1-150
