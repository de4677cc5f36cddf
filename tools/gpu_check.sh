#!/bin/bash
# Usage (on the GPU box via gpurun): tools/gpu_check.sh <tag> [bench args...]
# Runs the GPU parity tests, the benchmark, and a rocprofv3 kernel trace of a short benchmark run.
tag=$1; shift
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_$tag.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_$tag.log
python bench.py "$@" > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err; echo "bench rc=$?"
python - <<PY
import json
try:
    d = json.load(open("gpurun_out/bench_$tag.json"))
    print("value", d["value"], "tok/s  us/token", d["step_roofline"]["us_per_token_device"], " lm_head", d["roofline"]["avg_launch_us"], "us", d["roofline"]["achieved"], "GB/s loop", d["roofline"]["avg_launch_us_back_to_back_loop"])
    print("eager lo", d["step_roofline"]["per_kernel_class_us_eager_T_low"])
    print("eager hi", d["step_roofline"]["per_kernel_class_us_eager_T_high"])
    print("cpu", d.get("cpu_baseline"))
except Exception as e:
    print("bench parse failed", e); print(open("gpurun_out/bench_$tag.err").read()[-2000:])
PY
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d gpurun_out/prof_$tag -o trace -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/bench_prof_$tag.json 2> gpurun_out/bench_prof_$tag.err
db=$(find gpurun_out/prof_$tag -name "*.db" | head -1)
python tools/rocpd_stats.py $db gpurun_out/kernel_stats_$tag.md
rm -rf gpurun_out/prof_$tag
