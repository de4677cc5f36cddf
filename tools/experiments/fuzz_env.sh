#!/bin/bash
# The generate / session sweeps under random settings of the test-hook switches: the lock-step decode's fall-back paths (bit mask),
# decode steps per graph, workgroup cap of the persistent GEMM, staging arena size, prefetcher forced on / off.
s=${1:-1}
for i in $(seq 1 ${2:-16}); do
  s=$(( (s * 1103515245 + 12345) % 2147483648 ))
  off=$(( s % 64 )); gs=$(( 1 << ((s / 64) % 5) )); wgs=$(( (s / 512) % 3 == 0 ? 16 * (1 + (s / 2048) % 8) : 0 )); pf=$(( (s / 16384) % 3 ))
  envs="ZGPT2_DECODE_PATHS_OFF=$off ZGPT2_GRAPH_STEPS=$gs"
  [ $wgs -gt 0 ] && envs="$envs ZGPT2_GEMM_WGS=$wgs"
  [ $pf -lt 2 ] && envs="$envs ZGPT2_PREFETCH=$pf"
  r1=$(env $envs python tests/sweeps/generate.py $(( 30000 + i * 50 )) 30 2>&1 | tail -1)
  r2=$(env $envs python tests/sweeps/session.py $(( 30000 + i * 50 )) 20 2>&1 | tail -1)
  echo "$envs | $r1 | $r2"
done
