#!/bin/bash
out=gpurun_out/r4c; mkdir -p $out
B=tools/bin/gemm_bench
{
for abl in 0 1 2 4 8 12 15 16 32 48 64 112 127; do echo "== ov ABL=$abl"; ZGPT2_OV_ABL=$abl ZGPT2_GEMM_DBG=256 timeout 60 $B -k ov -stamps -nocheck -b 3 | grep -v "^check"; done
} > $out/gemm_c.txt 2>&1
cat $out/gemm_c.txt
