#!/bin/bash
for cfg in "2 0" "2 1" "3 0" "3 1"; do set -- $cfg; echo "== parts $1 prio-swap $2"; ZGPT2_DUAL_PARTS=$1 ZGPT2_DUAL_PRIO=$2 ZGPT2_TAG_SPIN_LIMIT=100000 timeout 300 python tools/dual_ab.py 124M 2>&1 | grep "dual=\|ids"; done
