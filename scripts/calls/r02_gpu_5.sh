#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for form in 1 2; do
  echo "=== form $form ==="
  HIOPAMD_DF_TILE=$form timeout -s KILL 120 python -u scripts/df_debug.py 2>&1 | tail -30
done
