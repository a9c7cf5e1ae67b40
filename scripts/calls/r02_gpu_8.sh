#!/bin/bash
# experiment: head tiles of the update follow the super-panel block row by block row (DF_UPH), A/B with the spine options
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "=== LDL^T / KKT parity tests (new defaults) ==="
timeout -s KILL 420 python -m pytest tests/test_gpu_ldlt_kkt.py -x -q > gpurun_out/pytest_ldlt.log 2>&1
echo "pytest exit: $?"; tail -5 gpurun_out/pytest_ldlt.log
for cfg in "1 7" "1 0" "1 1" "1 3" "0 7" "0 0"; do
  set -- $cfg
  echo "=== HIOPAMD_DF_UPH=$1 HIOPAMD_DF_SPINE=$2 ==="
  HIOPAMD_DF_UPH=$1 HIOPAMD_DF_SPINE=$2 DF_TIMELINE=0 timeout -s KILL 120 python scripts/df_stamps.py 2>&1 | tail -1
done
echo "=== stamps (UPH=1 SPINE=7) ==="
DF_MODES=1 timeout -s KILL 120 python scripts/df_stamps.py > gpurun_out/stamps8.log 2>&1; tail -22 gpurun_out/stamps8.log
