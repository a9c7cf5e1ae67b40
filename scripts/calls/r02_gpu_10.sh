#!/bin/bash
# experiment: look-ahead pivot reciprocal in the sub-block factor + chain role rebalancing; residual checked by df_stamps.py
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for rep in 1 2; do
for sp in 7 1 0; do
  echo "=== HIOPAMD_DF_SPINE=$sp (rep $rep) ==="
  HIOPAMD_DF_SPINE=$sp DF_TIMELINE=0 timeout -s KILL 120 python scripts/df_stamps.py 2>&1 | tail -1
done
done
echo "=== N = 4096, 2048 + 1000 (ragged), 8192 + 64 ==="
for n in 4096 3048 8256; do DF_N=$n DF_TIMELINE=0 timeout -s KILL 120 python scripts/df_stamps.py 2>&1 | tail -1; done
echo "=== stamps (defaults) ==="
DF_MODES=1 timeout -s KILL 120 python scripts/df_stamps.py > gpurun_out/stamps10.log 2>&1; tail -12 gpurun_out/stamps10.log
