for rep in 1 2; do
for v in old new; do
  ZGPT2_LIB=zig_gpt2_amd/lib/libzgpt2_hip_$v.so python bench.py --steps 5 --warmup 1 --no-cpu-baseline $@ 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])"
done; done
