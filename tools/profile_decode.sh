#!/bin/bash
# Usage (GPU box): tools/profile_decode.sh <tag> <model> [extra bench args]  -> gpurun_out/prof_<tag>/kernel_stats.md
tag=$1; model=$2; shift 2
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $out/trace -o bench -- python3 bench.py --model $model --steps 1 --warmup 1 --no-cpu-baseline "$@" > $out/bench_under_rocprof.json 2> $out/bench_under_rocprof.err
python tools/rocpd_stats.py $(find $out/trace -name "*.db" | head -1) $out/kernel_stats.md > /dev/null
rm -rf $out/trace
head -30 $out/kernel_stats.md
