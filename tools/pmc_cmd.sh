#!/bin/bash
# Usage (GPU box): tools/pmc_cmd.sh <tag> "<counters>" <kernel-name filter> python3 <script> [args]  -> gpurun_out/<tag>_pmc.md
# One rocprofv3 counter pass (--kernel-trace --pmc only) over any python3 command of this repo.
tag=$1; counters=$2; filt=$3; shift 3
out=gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --pmc $counters -d $out/pmc_$tag -o pmc -- "$@" > $out/${tag}_pmc.log 2> $out/${tag}_pmc.err
python tools/rocpd_pmc.py $(find $out/pmc_$tag -name "*.db" | head -1) 2> /dev/null | grep -E "$filt" | head -60 > $out/${tag}_pmc.md
rm -rf $out/pmc_$tag
cat $out/${tag}_pmc.md | cut -c1-200
