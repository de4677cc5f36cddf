#!/bin/bash
for r in 0 20 24 25 28 32 40 48 49 56 64 96 100 196; do
  echo -n "ZGPT2_RPW=$r: "; ZGPT2_RPW=$r python tools/kernel_chain.py 124M 1 2>/dev/null | grep lm_head
done
