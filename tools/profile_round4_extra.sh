#!/bin/bash
# Usage (GPU box): tools/profile_round4_extra.sh -> gpurun_out/profiles_round4x/*: counter passes the main script does not make —
# the 8-prompt decode with the 24-bit KV cache (FETCH / WRITE per kernel), the whole-prompt pass (MFMA busy, FETCH / WRITE of
# its GEMMs and of the causal attention at 1 x 1023 and 8 x 1023 tokens), the 768 x 3072 GEMM at M = 16384.
out=gpurun_out/profiles_round4x
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
pmc() {  # name, counters, program...
  local name=$1 c=$2; shift 2
  local n=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $out/p_${name}_$n -o pmc -- "$@" > $out/p_${name}_$n.log 2> $out/p_${name}_$n.err
  python tools/rocpd_pmc.py $(find $out/p_${name}_$n -name "*.db" | head -1) 2> /dev/null | head -40 >> $out/round4_${name}_pmc.md
  rm -rf $out/p_${name}_$n
}
for c in FETCH_SIZE WRITE_SIZE; do pmc 124m_8prompts_b24 $c python3 tools/pmc_decode.py 124M 8 b24; done
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" FETCH_SIZE WRITE_SIZE; do
  pmc prefill_1x1023 "$c" python3 tools/bench_prefill.py --batch 1 --lengths 1023 --reps 5
  pmc prefill_8x1023 "$c" python3 tools/bench_prefill.py --batch 8 --lengths 1023 --reps 5
  pmc gemm_m16384 "$c" python3 tools/bench_gemm.py 16384
done
ls -la $out
