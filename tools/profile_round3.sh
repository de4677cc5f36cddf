#!/bin/bash
# Usage (GPU box): tools/profile_round3.sh round3   -> gpurun_out/profiles_round3/* (copy what is judged into profiles/)
# Round 3: everything profile_round.sh collects for the one-prompt 124M step, plus kernel traces and FETCH / WRITE passes of
# the 8-prompt load and of GPT-2 XL, the GEMM counters of the four-wave kernel, and the bench lines of every BASELINE config.
tag=$1
out=gpurun_out/profiles_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
trace() {  # trace <name> <command...>: kernel trace + stats -> $out/${tag}_<name>_kernel_stats.md
  local name=$1; shift
  timeout 900 rocprofv3 --kernel-trace --stats -d $out/tr_$name -o t -- "$@" > $out/${name}_under_rocprof.json 2> $out/${name}_under_rocprof.err
  python tools/rocpd_stats.py $(find $out/tr_$name -name "*.db" | head -1) $out/${tag}_${name}_kernel_stats.md > /dev/null
  rm -rf $out/tr_$name
}
pmc() {  # pmc <name> <counter> <command...> -> $out/${tag}_<name>_pmc_<counter>.md (retried: rocprofv3 7.2 crashes now and then)
  local name=$1 c=$2; shift 2
  for attempt in 1 2 3; do
    timeout 900 rocprofv3 --kernel-trace --pmc $c -d $out/pmc_${name}_$c -o pmc -- "$@" > $out/pmc_${name}_$c.log 2> $out/pmc_${name}_$c.err
    python tools/rocpd_pmc.py $(find $out/pmc_${name}_$c -name "*.db" | head -1) $out/${tag}_${name}_pmc_$c.md.new > /dev/null 2> $out/pmc_${name}_${c}_parse.err
    rm -rf $out/pmc_${name}_$c
    if [ -s $out/${tag}_${name}_pmc_$c.md.new ]; then mv $out/${tag}_${name}_pmc_$c.md.new $out/${tag}_${name}_pmc_$c.md; break; fi
  done
}
# 1. kernel traces of the benchmark commands (full 1024-position context)
trace 124m python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
trace 124m_noprefetch python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prefetch
trace 124m_8prompts python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --prompts-per-gpu 8
trace xl python3 bench.py --model xl --steps 1 --warmup 1 --no-cpu-baseline
# 2. HBM traffic counters, separate passes, eager launches (tools/pmc_decode.py)
for c in FETCH_SIZE WRITE_SIZE; do
  pmc 124m $c python3 tools/pmc_decode.py 124M 1
  pmc 124m_8prompts $c python3 tools/pmc_decode.py 124M 8
  pmc xl $c python3 tools/pmc_decode.py xl 1
done
python tools/make_traffic_json.py $out $tag > /dev/null 2> $out/traffic.err
# 3. un-profiled lines of every BASELINE config (+ the two optional modes)
python bench.py --steps 5 --warmup 1 > $out/${tag}_bench.json 2> $out/bench.err
python bench.py --steps 5 --warmup 1 --no-prefetch --no-cpu-baseline > $out/${tag}_bench_noprefetch.json 2> $out/bench_noprefetch.err
python bench.py --steps 3 --warmup 1 --prompts-per-gpu 8 --no-cpu-baseline > $out/${tag}_bench_8prompts.json 2> $out/bench_8prompts.err
python bench.py --steps 3 --warmup 1 --prompts-per-gpu 8 --kv-f16 --no-cpu-baseline > $out/${tag}_bench_8prompts_kvf16.json 2> $out/bench_8prompts_kvf16.err
python bench.py --steps 2 --warmup 1 --model xl --no-cpu-baseline > $out/${tag}_bench_xl.json 2> $out/bench_xl.err
python bench.py --steps 5 --warmup 1 --model nano-char --no-cpu-baseline > $out/${tag}_bench_nano_char.json 2> $out/bench_nano_char.err
python bench.py --steps 3 --warmup 1 --weights-f32 --no-cpu-baseline > $out/${tag}_bench_weights_f32.json 2> $out/bench_weights_f32.err
# 4. the 768x3072 GEMM: kernel trace (durations) + counter passes, both kernel generations
for k in s4 p8; do
  ZGPT2_GEMM_KERNEL=$k trace gemm_$k python3 tools/bench_gemm.py 8192
  : > $out/${tag}_gemm_${k}_pmc.md
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
    n=$(echo $c | tr ' ' '_' | cut -c1-40)
    ZGPT2_GEMM_KERNEL=$k rocprofv3 --kernel-trace --pmc $c -d $out/pg_$n -o pmc -- python3 tools/bench_gemm.py 8192 > /dev/null 2> $out/pg_${k}_$n.err
    python tools/rocpd_pmc.py $(find $out/pg_$n -name "*.db" | head -1) 2>/dev/null | grep -i "gemm_\|^| kernel\|^|---" >> $out/${tag}_gemm_${k}_pmc.md
    rm -rf $out/pg_$n
  done
done
tools/bin/gemm_bench -k s4 > $out/${tag}_gemm_bench.txt 2>&1; tools/bin/gemm_bench -k p8 -nocheck >> $out/${tag}_gemm_bench.txt 2>&1
tools/bin/mfma_lds > $out/${tag}_mfma_lds.txt 2>&1; tools/bin/dma_intake > $out/${tag}_dma_intake.txt 2>&1
for a in "256 768 2000 256" "256 3072 2000 256" "64 768 2000 256"; do tools/bin/persistent_chain $a | tail -1; done > $out/${tag}_persistent_chain.txt 2>&1
# 4b. lock-step batch: planes / tagged hand-overs on and off, and the in-kernel timelines (stamps build)
python tools/planes_ab.py 124M:8 124M:3 2> /dev/null > $out/${tag}_planes_ab.jsonl
AB_KV_F16=1 python tools/planes_ab.py 124M:8 2> /dev/null >> $out/${tag}_planes_ab.jsonl
if [ -f zig_gpt2_amd/lib/libzgpt2_hip_stamps.so ]; then
  ZGPT2_LIB=zig_gpt2_amd/lib/libzgpt2_hip_stamps.so python tools/kernel_stamps.py 124M 1 2> /dev/null > $out/${tag}_kernel_stamps_1prompt.txt
  ZGPT2_LIB=zig_gpt2_amd/lib/libzgpt2_hip_stamps.so python tools/kernel_stamps.py 124M 8 2> /dev/null > $out/${tag}_kernel_stamps_8prompts.txt
fi
# 5. whole-prompt prefill: timings and kernel trace at 1023 prompt tokens
python tools/bench_prefill.py > $out/${tag}_prefill.jsonl 2> $out/prefill.err
python tools/bench_prefill.py --batch 8 --lengths 128,1023 >> $out/${tag}_prefill.jsonl 2>> $out/prefill.err
python tools/bench_prefill.py --planes 2 --lengths 128,1023 >> $out/${tag}_prefill.jsonl 2>> $out/prefill.err
python tools/bench_prefill.py --weights-f32 --lengths 128,1023 >> $out/${tag}_prefill.jsonl 2>> $out/prefill.err
trace prefill_1023 python3 tools/bench_prefill.py --lengths 1023 --reps 10
ls -la $out
