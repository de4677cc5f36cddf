#!/bin/bash
# Usage (GPU box): tools/pmc_prefill.sh <tag> <batch> "<counters>" [kernel-name filter] -> gpurun_out/<tag>_pmc.md
# One rocprofv3 counter pass (--kernel-trace --pmc only) over the whole-prompt pass of tools/bench_prefill.py.
tag=$1; batch=$2; counters=$3; filt=${4:-.}
out=gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --pmc $counters -d $out/pmc_$tag -o pmc -- python3 tools/bench_prefill.py --batch $batch --lengths 1023 --reps 3 > $out/${tag}_pmc.log 2> $out/${tag}_pmc.err
python tools/rocpd_pmc.py $(find $out/pmc_$tag -name "*.db" | head -1) 2> /dev/null | grep -E "$filt" | head -60 > $out/${tag}_pmc.md
rm -rf $out/pmc_$tag
cat $out/${tag}_pmc.md | cut -c1-160
