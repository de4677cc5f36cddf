#!/bin/bash
# usage: [env ...] tools/prof_gemm2.sh <tag> <M> [N K]   — two PMC passes (MFMA busy / clock, wait + LDS counters) of bench_gemm.py
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $c -d gpurun_out/pg_$n -o pmc -- python3 tools/bench_gemm.py "$@" > /dev/null 2> gpurun_out/pg_${tag}_$n.err
  python tools/rocpd_pmc.py $(find gpurun_out/pg_$n -name "*.db" | head -1) 2>/dev/null | grep -i "gemm" | awk -F'|' -v t=$tag '{print t, $2, $3, $4, $5}' | sed 's/(unsigned short.*voi`//'
  rm -rf gpurun_out/pg_$n
done
