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
# 3. un-profiled reference lines: the headline config and the other BASELINE configs
python bench.py --steps 5 --warmup 1 > $out/bench.json 2> $out/bench.err
python bench.py --steps 3 --warmup 1 --prompts-per-gpu 8 --no-cpu-baseline > $out/bench_8prompts.json 2> $out/bench_8prompts.err
python bench.py --steps 2 --warmup 1 --model xl --no-cpu-baseline > $out/bench_xl.json 2> $out/bench_xl.err
python bench.py --steps 5 --warmup 1 --model nano-char --no-cpu-baseline > $out/bench_nano_char.json 2> $out/bench_nano_char.err
# 4. whole-prompt prefill: timings and kernel trace at 1023 and 128 prompt tokens
export PYTHONPATH=$GRAFT_REPO_ROOT
python tools/bench_prefill.py > $out/prefill.jsonl 2> $out/prefill.err
python tools/bench_prefill.py --batch 8 --lengths 128,1023 >> $out/prefill.jsonl 2>> $out/prefill.err
for n in 1023 128; do
  rocprofv3 --kernel-trace --stats -d $out/pf_$n -o pf -- python3 tools/bench_prefill.py --lengths $n --reps 10 > /dev/null 2> $out/pf_$n.err
  python tools/rocpd_stats.py $(find $out/pf_$n -name "*.db" | head -1) $out/prefill_${n}_kernel_stats.md > /dev/null
  rm -rf $out/pf_$n
done
ls -la $out
