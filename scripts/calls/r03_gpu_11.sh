#!/bin/bash
# round 3, GPU call 11: K = 512 pairing depth with the round-2 queue order (HIOPAMD_DF_SPLIT=0), full V workspaces
set -u
export TMPDIR=/tmp
for cfg in "HIOPAMD_DF_K512=0" "HIOPAMD_DF_K512=8" "HIOPAMD_DF_K512=12" "HIOPAMD_DF_K512=16" "HIOPAMD_DF_K512=20" "HIOPAMD_DF_K512=24" "HIOPAMD_DF_K512=16 HIOPAMD_DF_NVB=4" "HIOPAMD_DF_K512=0" "HIOPAMD_DF_K512=16"; do
  echo "=== SPLIT=0 $cfg"
  env HIOPAMD_DF_SPLIT=0 $cfg DF_TIMELINE=0 timeout -s KILL 180 python scripts/df_stamps.py 2>&1 | tail -1
done
