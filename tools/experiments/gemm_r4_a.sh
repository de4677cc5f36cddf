#!/bin/bash
# round 4, experiment batch A (GPU box): where the 768x3072 GEMM's time is — wall-clock stamps, store cache policies, band width
out=gpurun_out/r4a; mkdir -p $out
B=tools/bin/gemm_bench
{
echo "== baseline + stamps"; ZGPT2_GEMM_DBG=256 $B -k s4 -stamps
echo "== no stamps";        $B -k s4 -nocheck
for abl in 16 32 48; do echo "== store policy ABL=$abl"; ZGPT2_S4_ABL=$abl ZGPT2_GEMM_DBG=256 $B -k s4 -stamps; done
echo "== no global stores (dbg 1)"; ZGPT2_GEMM_DBG=257 $B -k s4 -stamps -nocheck
echo "== no epilogue (dbg 4)";      ZGPT2_GEMM_DBG=260 $B -k s4 -stamps -nocheck
for gw in 1 2 4 16; do echo "== band width $gw"; ZGPT2_GW=$gw ZGPT2_GEMM_DBG=256 $B -k s4 -stamps -nocheck; done
echo "== no gelu"; ZGPT2_GEMM_DBG=256 $B -k s4 -stamps -nogelu -nocheck
echo "== f32 out no gelu"; ZGPT2_GEMM_DBG=256 $B -k s4 -stamps -nogelu -f32 -nocheck
echo "== zero fill"; ZGPT2_GEMM_DBG=256 $B -k s4 -stamps -fill 1 -nocheck
echo "== M=16384"; ZGPT2_GEMM_DBG=256 $B 16384 3072 768 -k s4 -stamps -nocheck
echo "== M=1024"; ZGPT2_GEMM_DBG=256 $B 1024 3072 768 -k s4 -stamps -nocheck
} > $out/gemm_a.txt 2>&1
python bench.py --steps 3 --warmup 1 > $out/bench.json 2> $out/bench.err
tail -5 $out/bench.err
cat $out/gemm_a.txt
