#!/bin/bash
out=gpurun_out/r4d; mkdir -p $out
B=tools/bin/gemm_bench
{
for abl in 0 32 128 160; do echo "== ov ABL=$abl (sc1 stores)"; ZGPT2_OV_ABL=$abl ZGPT2_GEMM_DBG=256 timeout 60 $B -k ov -stamps -nocheck -b 3 | grep -v "^check\|^stamps: 256"; done
for abl in 0 1 2 4 7; do echo "== s4 ABL=$abl"; ZGPT2_S4_ABL=$abl ZGPT2_GEMM_DBG=256 timeout 60 $B -k s4 -stamps -nocheck -b 3 | grep -v "^check\|^stamps: 256"; done
} > $out/gemm_d.txt 2>&1
cat $out/gemm_d.txt
