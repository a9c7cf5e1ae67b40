#!/bin/bash
# round 3, GPU call 21: one workgroup per CU (the one-dispatch form) against two — what the second resident workgroup buys
set -u
export TMPDIR=/tmp
for cfg in "HIOPAMD_DF_ONE=0" "HIOPAMD_DF_ONE=1" "HIOPAMD_DF_ONE=1 HIOPAMD_DF_K512=0" "HIOPAMD_DF_ONE=0 HIOPAMD_DF_K512=0"; do
  echo "=== $cfg"
  env $cfg DF_TIMELINE=0 timeout -s KILL 180 python scripts/df_stamps.py 2>&1 | tail -1
done
