#!/bin/bash
# Usage (GPU box): tools/profile_round.sh round2   -> gpurun_out/profiles_round2/* (copy what is judged into profiles/)
tag=$1
out=gpurun_out/profiles_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
# 1. kernel trace + stats of the default benchmark command (full 1024-position context)
timeout 600 rocprofv3 --kernel-trace --stats -d $out/trace -o bench -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/bench_under_rocprof.err
python tools/rocpd_stats.py $(find $out/trace -name "*.db" | head -1) $out/${tag}_kernel_stats.md > /dev/null
rm -rf $out/trace
# 1b. the same without the side-stream L2 prefetcher: the per-kernel durations it changes
timeout 600 rocprofv3 --kernel-trace --stats -d $out/trace -o bench -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prefetch > $out/bench_under_rocprof_noprefetch.json 2> $out/bench_under_rocprof_noprefetch.err
python tools/rocpd_stats.py $(find $out/trace -name "*.db" | head -1) $out/${tag}_kernel_stats_noprefetch.md > /dev/null
rm -rf $out/trace
# 2. HBM traffic counters of the same workload at the full 1024-position context, separate passes.  Eager launches
#    (--no-graph: the same kernels without hipGraph replay) — rocprofv3 7.2 segfaults collecting counters over
#    graph replays of more than a few hundred steps, and intermittently behind long queues: tools/pmc_decode.py is the
#    leanest target (one generation, no prefetcher: every kernel's own traffic), retried when the tool crashes.
for c in FETCH_SIZE WRITE_SIZE; do
  for attempt in 1 2 3; do
    timeout 600 rocprofv3 --kernel-trace --pmc $c -d $out/pmc_$c -o pmc -- python3 tools/pmc_decode.py > $out/pmc_$c.log 2> $out/pmc_$c.err
    python tools/rocpd_pmc.py $(find $out/pmc_$c -name "*.db" | head -1) $out/${tag}_pmc_$c.md.new > /dev/null 2> $out/pmc_${c}_parse.err
    rm -rf $out/pmc_$c
    # a crashed pass leaves no empty table behind; try again
    if [ -s $out/${tag}_pmc_$c.md.new ]; then mv $out/${tag}_pmc_$c.md.new $out/${tag}_pmc_$c.md; break; fi
  done
done
python tools/make_traffic_json.py $out $tag > /dev/null 2> $out/traffic.err
# 3. un-profiled lines: the headline config and the other BASELINE configs
python bench.py --steps 5 --warmup 1 > $out/${tag}_bench.json 2> $out/bench.err
python bench.py --steps 5 --warmup 1 --no-prefetch --no-cpu-baseline > $out/${tag}_bench_noprefetch.json 2> $out/bench_noprefetch.err
python bench.py --steps 3 --warmup 1 --prompts-per-gpu 8 --no-cpu-baseline > $out/${tag}_bench_8prompts.json 2> $out/bench_8prompts.err
python bench.py --steps 2 --warmup 1 --model xl --no-cpu-baseline > $out/${tag}_bench_xl.json 2> $out/bench_xl.err
python bench.py --steps 5 --warmup 1 --model nano-char --no-cpu-baseline > $out/${tag}_bench_nano_char.json 2> $out/bench_nano_char.err
# 4. the 768x3072 GEMM: kernel trace (durations) + counter passes
rocprofv3 --kernel-trace --stats -d $out/gtrace -o g -- python3 tools/bench_gemm.py 8192 > $out/gemm_under_rocprof.json 2> $out/gemm_under_rocprof.err
python tools/rocpd_stats.py $(find $out/gtrace -name "*.db" | head -1) $out/${tag}_gemm_kernel_stats.md > /dev/null
rm -rf $out/gtrace
: > $out/${tag}_gemm_pmc.md
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  n=$(echo $c | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $c -d $out/pg_$n -o pmc -- python3 tools/bench_gemm.py 8192 > /dev/null 2> $out/pg_$n.err
  python tools/rocpd_pmc.py $(find $out/pg_$n -name "*.db" | head -1) 2>/dev/null | grep -i "gemm_p8\|^| kernel\|^|---" >> $out/${tag}_gemm_pmc.md
  rm -rf $out/pg_$n
done
# 5. GPT-2 XL: kernel trace
rocprofv3 --kernel-trace --stats -d $out/xtrace -o x -- python3 bench.py --model xl --steps 1 --warmup 1 --no-cpu-baseline > $out/xl_under_rocprof.json 2> $out/xl_under_rocprof.err
python tools/rocpd_stats.py $(find $out/xtrace -name "*.db" | head -1) $out/${tag}_xl_kernel_stats.md > /dev/null
rm -rf $out/xtrace
# 6. whole-prompt prefill: timings and kernel trace at 1023 prompt tokens
python tools/bench_prefill.py > $out/${tag}_prefill.jsonl 2> $out/prefill.err
python tools/bench_prefill.py --batch 8 --lengths 128,1023 >> $out/${tag}_prefill.jsonl 2>> $out/prefill.err
python tools/bench_prefill.py --planes 2 --lengths 128,1023 >> $out/${tag}_prefill.jsonl 2>> $out/prefill.err
python tools/bench_prefill.py --planes 2 --batch 8 --lengths 1023 >> $out/${tag}_prefill.jsonl 2>> $out/prefill.err
python tools/bench_prefill.py --weights-f32 --lengths 128,1023 >> $out/${tag}_prefill.jsonl 2>> $out/prefill.err
rocprofv3 --kernel-trace --stats -d $out/pf -o pf -- python3 tools/bench_prefill.py --lengths 1023 --reps 10 > /dev/null 2> $out/pf.err
python tools/rocpd_stats.py $(find $out/pf -name "*.db" | head -1) $out/${tag}_prefill_1023_kernel_stats.md > /dev/null
rm -rf $out/pf
ls -la $out
