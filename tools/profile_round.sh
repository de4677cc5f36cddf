#!/bin/bash
# Usage (GPU box): tools/profile_round.sh r01   -> gpurun_out/profiles_r01/*.md (copy into profiles/)
tag=$1
out=gpurun_out/profiles_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# 1. kernel trace + stats of the default benchmark command
rocprofv3 --kernel-trace --stats -d $out/trace -o bench -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/bench_under_rocprof.err
python tools/rocpd_stats.py $(find $out/trace -name "*.db" | head -1) $out/kernel_stats.md > /dev/null
rm -rf $out/trace
# 2. HBM traffic counters, separate passes, short context to bound the serialised-dispatch run time
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $out/pmc_$c -o pmc -- python3 bench.py --steps 1 --warmup 0 --ctx 64 --no-cpu-baseline > $out/pmc_$c.json 2> $out/pmc_$c.err
  python tools/rocpd_pmc.py $(find $out/pmc_$c -name "*.db" | head -1) $out/pmc_$c.md > /dev/null 2> $out/pmc_${c}_parse.err
  rm -rf $out/pmc_$c
done
# 3. un-profiled reference line
python bench.py --steps 5 --warmup 1 > $out/bench.json 2> $out/bench.err
ls -la $out
