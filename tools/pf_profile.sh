cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for mode in 2 0; do
  ZGPT2_PF_MODE=$mode ZGPT2_PF_XSHIFT=1 ZGPT2_PF_NSUB=8 ZGPT2_PF_CLASSES=62 rocprofv3 --kernel-trace --stats -d gpurun_out/pf_mode$mode -o pf -- python3 tools/pf_once.py 124M > gpurun_out/pf_mode$mode.log 2>&1
  f=$(find gpurun_out/pf_mode$mode -name "*kernel_stats.csv" | head -1)
  echo "== mode $mode"; cat gpurun_out/pf_mode$mode.log | grep us/token; head -9 $f | cut -d, -f1-8 | cut -c1-200
done
