#!/bin/bash
# zgpt2_main with arguments it cannot serve: every invocation must end with a message and a non-zero status, not a signal.
B=zig_gpt2_amd/bin/zgpt2_main
run() { timeout 60 "$@" > /tmp/h.out 2>&1; rc=$?; printf "%-70s -> rc %3d  %s\n" "$*" $rc "$(tail -1 /tmp/h.out | cut -c1-90)"; if [ $rc -ge 124 ]; then echo "   ^^^ signal / timeout"; fi; }
run $B
run $B tiny
run $B nosuchmodel 1 1,2,3 10
run $B tiny 1 "" 10
run $B tiny 1 1,2,3 0
run $B tiny 1 1,2,3 100000
run $B tiny 1 1,2,999999 10
run $B tiny 1 1,-2,3 10
run $B tiny 1 a,b 10
run $B tiny /no/such/dir 1,2,3 10
run $B tiny 1 1,2,3 10 --nosuchflag
run $B tiny 1 1,2,3 10 --gpus 0
run $B tiny 1 "1,2,3;4,5;6" 10 --gpus 3
run $B tiny 1 "1,2,3;4,5" 10 --gpus 1
run $B tiny 1 1,2,3 10 --gpus 99
run $B tiny 1 1,2,3 10 --gpus 1
run $B tiny 1 1,2,3 10 --model-tier
run $B tiny 1 1,2,3 64
run $B tiny 1 1,2,3 65
