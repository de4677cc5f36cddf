#!/bin/bash
# (round 4's sweep also ran a three-stream variant and swapped stream priorities: profiles/round4_dual_decode_sweep*.txt; that
# variant lost — 218.6-220.0 against 213.2-214.7 us per token — and was removed)
ZGPT2_TAG_SPIN_LIMIT=100000 timeout 300 python tools/dual_ab.py 124M 2>&1 | grep "dual=\|ids"
