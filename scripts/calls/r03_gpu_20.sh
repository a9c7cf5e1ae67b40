#!/bin/bash
# round 3, GPU call 20: NEAR/FAR queues together with one row-panel workspace per super-panel (the combination r03_gpu_10 did not run)
set -u
mkdir -p gpurun_out/r03_20
export TMPDIR=/tmp
O=gpurun_out/r03_20
for cfg in "HIOPAMD_DF_SPLIT=0" "HIOPAMD_DF_SPLIT=1 HIOPAMD_DF_NVB=32" "HIOPAMD_DF_SPLIT=1 HIOPAMD_DF_NVB=8" "HIOPAMD_DF_SPLIT=1 HIOPAMD_DF_NVB=32 HIOPAMD_DF_K512=24" "HIOPAMD_DF_SPLIT=1 HIOPAMD_DF_NVB=32 HIOPAMD_DF_K512=1073741824" "HIOPAMD_DF_SPLIT=1 HIOPAMD_DF_NVB=32 HIOPAMD_DF_K512=0" "HIOPAMD_DF_SPLIT=0 HIOPAMD_DF_NVB=32" "HIOPAMD_DF_SPLIT=0"; do
  echo "=== $cfg"
  env $cfg DF_TIMELINE=0 timeout -s KILL 180 python scripts/df_stamps.py 2>&1 | tail -1
done
HIOPAMD_DF_SPLIT=1 HIOPAMD_DF_NVB=32 DF_TIMELINE=1 DF_MODES=1 timeout -s KILL 180 python scripts/df_stamps.py > $O/timeline_split_nvb32.txt 2>&1
head -36 $O/timeline_split_nvb32.txt | tail -34
