#!/bin/bash
# Usage (GPU box): tools/profile_round6.sh round6 [parts]  -> gpurun_out/profiles_round6/* (copy what is judged into profiles/)
# parts (default all): bench trace traffic gemm corun optier
# Round 6: the default bench line; kernel trace of the one-prompt 124M step; FETCH / WRITE counter passes -> traffic.json; the
# GEMM account (c_fc bias + GELU, bias only, the mlp c_proj orientation): trace + MFMA-busy counters; co-running prompt groups:
# kernel trace + overlap analysis; the op tier under the C++ host: kernel trace.
tag=$1; parts=${2:-"bench trace traffic gemm corun optier"}
out=gpurun_out/profiles_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
has() { [[ " $parts " == *" $1 "* ]]; }
trace() {
  local name=$1; shift
  timeout 900 rocprofv3 --kernel-trace --stats -d $out/tr_$name -o t -- "$@" > $out/${name}_under_rocprof.out 2> $out/${name}_under_rocprof.err
  python tools/rocpd_stats.py $(find $out/tr_$name -name "*.db" | head -1) $out/${tag}_${name}_kernel_stats.md > /dev/null
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
if has bench; then
  python bench.py --steps 5 --warmup 1 > $out/${tag}_bench.json 2> $out/bench.err
  python bench.py --steps 3 --warmup 1 --prompts-per-gpu 8 --no-cpu-baseline --no-op-tier > $out/${tag}_bench_8prompts.json 2> $out/bench_8prompts.err
fi
if has trace; then
  trace 124m python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-op-tier
  rm -rf $out/tr_124m
fi
if has traffic; then
  for c in FETCH_SIZE WRITE_SIZE; do
    pmc 124m $c python3 tools/pmc_decode.py 124M 1
    pmc 124m_8prompts $c python3 tools/pmc_decode.py 124M 8
    pmc xl $c python3 tools/pmc_decode.py xl 1
  done
  trace 124m_8prompts python3 bench.py --steps 2 --warmup 1 --prompts-per-gpu 8 --no-cpu-baseline --no-op-tier; rm -rf $out/tr_124m_8prompts
  trace xl python3 bench.py --model xl --steps 1 --warmup 1 --no-cpu-baseline --no-op-tier; rm -rf $out/tr_xl
  python tools/make_traffic_json.py $out $tag > /dev/null 2> $out/traffic.err
fi
if has gemm; then
  : > $out/${tag}_gemm_account_pmc.md
  for v in "8192 3072 768" "8192 3072 768 nogelu" "16384 768 3072 nogelu" "8192 768 3072 nogelu"; do
    n=$(echo $v | tr ' ' '_')
    echo "## bench_gemm.py $v" >> $out/${tag}_gemm_account_pmc.md
    python3 tools/bench_gemm.py $v 2>/dev/null | tail -1 >> $out/${tag}_gemm_account_pmc.md
    for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
      cn=$(echo $c | tr ' ' '_' | cut -c1-40)
      timeout 600 rocprofv3 --kernel-trace --pmc $c -d $out/pg_$cn -o pmc -- python3 tools/bench_gemm.py $v > /dev/null 2> $out/pg_${n}_$cn.err
      python tools/rocpd_pmc.py $(find $out/pg_$cn -name "*.db" | head -1) 2>/dev/null | grep -i "gemm_\|^| kernel\|^|---" >> $out/${tag}_gemm_account_pmc.md
      rm -rf $out/pg_$cn
    done
    timeout 600 rocprofv3 --kernel-trace --stats -d $out/tg_$n -o t -- python3 tools/bench_gemm.py $v > /dev/null 2> $out/tg_$n.err
    python tools/rocpd_stats.py $(find $out/tg_$n -name "*.db" | head -1) 2>/dev/null | grep -i "gemm_" >> $out/${tag}_gemm_account_pmc.md
    rm -rf $out/tg_$n
  done
  { ZGPT2_GEMM_DBG=256 tools/bin/gemm_bench -k s4 -stamps; ZGPT2_GEMM_DBG=256 tools/bin/gemm_bench 8192 3072 768 -k s4 -stamps -nogelu -nocheck; ZGPT2_GEMM_DBG=256 tools/bin/gemm_bench 16384 768 3072 -k s4 -stamps -nogelu -nocheck; } > $out/${tag}_gemm_bench.txt 2>&1
fi
if has corun; then
  for g in 1 2 4; do
    timeout 600 rocprofv3 --kernel-trace -d $out/tc_$g -o t -- python3 tools/experiments/corun_ab.py --ctx 256 --gens 1 --prio normal $g > $out/corun_$g.out 2> $out/corun_$g.err
    python tools/rocpd_overlap.py $(find $out/tc_$g -name "*.db" | head -1) $out/${tag}_corun_G${g}_overlap.md "gemv|attn|embed|lm_head" > /dev/null
    rm -rf $out/tc_$g
  done
  for prio in cycle normal; do python3 tools/experiments/corun_ab.py --prio $prio 1 2 4 8 2>/dev/null | grep "^{"; done > $out/${tag}_corun_ab.jsonl
fi
if has optier; then
  trace optier ./zig_gpt2_amd/bin/zgpt2_main 124M 0 1000 256
  rm -rf $out/tr_optier
  { ./zig_gpt2_amd/bin/zgpt2_main 124M 0 1000 64 > /dev/null; ./zig_gpt2_amd/bin/zgpt2_main 124M 0 1000 1024 > /dev/null; ZGPT2_OP_POLL=0 ./zig_gpt2_amd/bin/zgpt2_main 124M 0 1000 1024 > /dev/null; } 2> $out/${tag}_op_tier.txt
  python3 tools/experiments/op_tier_calls.py 512 > $out/${tag}_op_tier_calls.json 2>/dev/null
fi
ls -la $out
