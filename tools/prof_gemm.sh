#!/bin/bash
# usage: tools/prof_gemm.sh <M> [N K]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  n=$(echo $c | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $c -d gpurun_out/pg_$n -o pmc -- python3 tools/bench_gemm.py "$@" > /dev/null 2> gpurun_out/pg_$n.err
  python tools/rocpd_pmc.py $(find gpurun_out/pg_$n -name "*.db" | head -1) 2>/dev/null | grep -i "gemm" | awk -F'|' '{print $3, $4, $5}'
  rm -rf gpurun_out/pg_$n
done
