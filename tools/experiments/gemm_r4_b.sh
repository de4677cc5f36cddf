#!/bin/bash
# round 4, batch B: the overlapped-epilogue kernel (gemm_ov) — correctness against the fp64 sample and bitwise against s4, timing, stamps
out=gpurun_out/r4b; mkdir -p $out
B=tools/bin/gemm_bench
{
for shape in "256 192 128" "512 384 256" "1100 776 320" "8192 3072 768"; do
  echo "== ov $shape gelu";   timeout 60 $B $shape -k ov -ref s4 -b 1 -i 5
  echo "== ov $shape nogelu"; timeout 60 $B $shape -k ov -ref s4 -nogelu -b 1 -i 5
done
echo "== few workgroups (many tiles each)"; ZGPT2_GEMM_WGS=3 timeout 60 $B 1100 776 320 -k ov -ref s4 -b 1 -i 5
ZGPT2_GEMM_WGS=5 timeout 60 $B 2048 1536 768 -k ov -ref s4 -b 1 -i 5
echo "== timing ov sc1"; ZGPT2_GEMM_DBG=256 timeout 60 $B -k ov -stamps
echo "== timing ov plain stores"; ZGPT2_OV_AUX=0 ZGPT2_GEMM_DBG=256 timeout 60 $B -k ov -stamps -nocheck
echo "== timing ov no stamps"; timeout 60 $B -k ov -nocheck
echo "== timing s4"; ZGPT2_GEMM_DBG=256 timeout 60 $B -k s4 -stamps -nocheck
echo "== ov no global stores"; ZGPT2_GEMM_DBG=257 timeout 60 $B -k ov -stamps -nocheck
echo "== ov nogelu"; ZGPT2_GEMM_DBG=256 timeout 60 $B -k ov -stamps -nogelu -nocheck
echo "== ov M=16384"; ZGPT2_GEMM_DBG=256 timeout 60 $B 16384 3072 768 -k ov -stamps -nocheck
echo "== ov zero fill"; ZGPT2_GEMM_DBG=256 timeout 60 $B -k ov -stamps -fill 1 -nocheck
} > $out/gemm_b.txt 2>&1
cat $out/gemm_b.txt
