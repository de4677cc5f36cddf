#!/bin/bash
# Usage (GPU box): tools/profile_round5.sh round5   -> gpurun_out/profiles_round5/* (copy what is judged into profiles/)
# Round 4: kernel trace + FETCH / WRITE passes of the one-prompt 124M step (traffic.json), the bench lines (default with its
# other_configs block, 8 prompts, 8 prompts with the fp16 cache, fp32 weights), the GEMM point (write-through stores): trace +
# counter passes, gemm_bench with stamps, prefill timings.
tag=$1
out=gpurun_out/profiles_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
trace() {
  local name=$1; shift
  timeout 900 rocprofv3 --kernel-trace --stats -d $out/tr_$name -o t -- "$@" > $out/${name}_under_rocprof.json 2> $out/${name}_under_rocprof.err
  python tools/rocpd_stats.py $(find $out/tr_$name -name "*.db" | head -1) $out/${tag}_${name}_kernel_stats.md > /dev/null
  rm -rf $out/tr_$name
}
pmc() {
  local name=$1 c=$2; shift 2
  for attempt in 1 2 3; do
    timeout 900 rocprofv3 --kernel-trace --pmc $c -d $out/pmc_${name}_$c -o pmc -- "$@" > $out/pmc_${name}_$c.log 2> $out/pmc_${name}_$c.err
    python tools/rocpd_pmc.py $(find $out/pmc_${name}_$c -name "*.db" | head -1) $out/${tag}_${name}_pmc_$c.md.new > /dev/null 2> $out/pmc_${name}_${c}_parse.err
    rm -rf $out/pmc_${name}_$c
    if [ -s $out/${tag}_${name}_pmc_$c.md.new ]; then mv $out/${tag}_${name}_pmc_$c.md.new $out/${tag}_${name}_pmc_$c.md; break; fi
  done
}
python bench.py --steps 5 --warmup 1 > $out/${tag}_bench.json 2> $out/bench.err
python bench.py --steps 3 --warmup 1 --prompts-per-gpu 8 --no-cpu-baseline > $out/${tag}_bench_8prompts.json 2> $out/bench_8prompts.err
python bench.py --steps 3 --warmup 1 --prompts-per-gpu 8 --kv-f16 --no-cpu-baseline > $out/${tag}_bench_8prompts_kvf16.json 2> $out/bench_8prompts_kvf16.err
python bench.py --steps 3 --warmup 1 --prompts-per-gpu 8 --kv-b24 --no-cpu-baseline > $out/${tag}_bench_8prompts_kvb24.json 2> $out/bench_8prompts_kvb24.err
python bench.py --steps 3 --warmup 1 --weights-f32 --no-cpu-baseline > $out/${tag}_bench_weights_f32.json 2> $out/bench_weights_f32.err
trace 124m python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs
for c in FETCH_SIZE WRITE_SIZE; do
  pmc 124m $c python3 tools/pmc_decode.py 124M 1
  pmc 124m_8prompts $c python3 tools/pmc_decode.py 124M 8
  pmc xl $c python3 tools/pmc_decode.py xl 1
done
trace 124m_8prompts python3 bench.py --steps 2 --warmup 1 --prompts-per-gpu 8 --no-cpu-baseline
trace xl python3 bench.py --model xl --steps 1 --warmup 1 --no-cpu-baseline
python tools/make_traffic_json.py $out $tag > /dev/null 2> $out/traffic.err
ZGPT2_GEMM_KERNEL=s4 trace gemm_s4 python3 tools/bench_gemm.py 8192
: > $out/${tag}_gemm_s4_pmc.md
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-40)
  ZGPT2_GEMM_KERNEL=s4 rocprofv3 --kernel-trace --pmc $c -d $out/pg_$n -o pmc -- python3 tools/bench_gemm.py 8192 > /dev/null 2> $out/pg_s4_$n.err
  python tools/rocpd_pmc.py $(find $out/pg_$n -name "*.db" | head -1) 2>/dev/null | grep -i "gemm_\|^| kernel\|^|---" >> $out/${tag}_gemm_s4_pmc.md
  rm -rf $out/pg_$n
done
{ ZGPT2_GEMM_DBG=256 tools/bin/gemm_bench -k s4 -stamps; tools/bin/gemm_bench -k p8 -nocheck; ZGPT2_GEMM_DBG=256 tools/bin/gemm_bench 16384 3072 768 -k s4 -stamps -nocheck; } > $out/${tag}_gemm_bench.txt 2>&1
for m in 1024 4096 8192 12288 16384 32768; do python tools/bench_gemm.py $m 2>/dev/null | tail -1; done > $out/${tag}_gemm_by_m.jsonl
python tools/bench_prefill.py > $out/${tag}_prefill.jsonl 2> $out/prefill.err
python tools/bench_prefill.py --batch 8 --lengths 128,1023 >> $out/${tag}_prefill.jsonl 2>> $out/prefill.err
python tools/bench_prefill.py --planes 2 --lengths 128,1023 >> $out/${tag}_prefill.jsonl 2>> $out/prefill.err
python tools/bench_prefill.py --weights-f32 --lengths 128,1023 >> $out/${tag}_prefill.jsonl 2>> $out/prefill.err
ls -la $out
# round 5: the whole-prompt pass (kernel traces at 1 x 1023 and 8 x 1023 tokens; MFMA busy, FETCH / WRITE of its kernels), the
# prompt Linears one by one on both GEMM families, the hardware-rule probes
for b in 1 8; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $out/tr_pf$b -o t -- python3 tools/bench_prefill.py --batch $b --lengths 1023 --reps 10 > /dev/null 2> $out/pf$b.err
  python tools/rocpd_stats.py $(find $out/tr_pf$b -name "*.db" | head -1) $out/${tag}_prefill_${b}x1023_kernel_stats.md > /dev/null
  rm -rf $out/tr_pf$b
  : > $out/${tag}_prefill_${b}x1023_pmc.md
  for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"; do
    n=$(echo $c | tr ' ' '_' | cut -c1-40)
    timeout 600 rocprofv3 --kernel-trace --pmc $c -d $out/pp_$n -o pmc -- python3 tools/bench_prefill.py --batch $b --lengths 1023 --reps 5 > /dev/null 2> $out/pp_$n.err
    python tools/rocpd_pmc.py $(find $out/pp_$n -name "*.db" | head -1) 2> /dev/null | grep -E "gemm|attn_prefill|reduce|ln_split|^\| kernel|^\|---" >> $out/${tag}_prefill_${b}x1023_pmc.md
    rm -rf $out/pp_$n
  done
done
{ for a in "8184 3072 768 2 1" "8184 3072 768 2 2" "8184 768 3072 1 1 2" "8184 768 3072 1 2" "8184 768 768 1 1 2" "8184 768 768 1 2"; do python tools/bench_prefill_linear.py $a 2>/dev/null | tail -1; done; } > $out/${tag}_prefill_linears.jsonl
{ tools/bin/tr_read_probe; tools/bin/soffset_bounds_probe; } > $out/${tag}_hw_rule_probes.txt 2>&1
{ for a in "8192 3072 768" "8192 3072 2304" "8192 4096 4096"; do tools/bin/gemm_bench $a -k s4 -nocheck 2>&1 | grep "^time"; done; tools/bin/gemm_bench 8192 3072 2304 -k s4 -nocheck -fill 1 2>&1 | grep "^time"; tools/bin/gemm_bench 8192 3072 768 -k s4 -nocheck -nogelu 2>&1 | grep "^time"; } > $out/${tag}_gemm_shapes.txt
ls -la $out
# round 5, second half: the prompt attention alone (timing, counters, workgroup timeline from the stamps build), the fp32-weight
# pass at 8 prompts, the epilogue / ping-pong microbenchmarks
{ for b in 8 4 2 1; do python3 tools/bench_attn_prefill.py $b 1023 12 2>/dev/null | tail -1; done; } > $out/${tag}_attn_prefill.jsonl
python tools/bench_prefill.py --weights-f32 --batch 8 --lengths 1023 >> $out/${tag}_prefill.jsonl 2>> $out/prefill.err
python tools/bench_prefill.py --batch 4 --lengths 1023 >> $out/${tag}_prefill.jsonl 2>> $out/prefill.err
if [ -f zig_gpt2_amd/lib/libzgpt2_hip_stamps.so ]; then ZGPT2_LIB=zig_gpt2_amd/lib/libzgpt2_hip_stamps.so python3 tools/attn_timeline.py 8 1023 12 > $out/${tag}_attn_timeline_8x1023.txt 2>/dev/null; fi
tools/bin/epilogue_issue > $out/${tag}_epilogue_issue.txt 2>&1
tools/bin/pingpong_probe > $out/${tag}_pingpong_probe.txt 2>&1
ls -la $out
