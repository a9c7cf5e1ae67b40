#!/bin/bash
# experiment: MFMA rank-1 form of the 16 x 16 sub-block factor + spine latency cuts (HIOPAMD_DF_SPINE bits), A/B in one call
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "=== LDL^T / KKT parity tests (new defaults) ==="
timeout -s KILL 420 python -m pytest tests/test_gpu_ldlt_kkt.py -x -q > gpurun_out/pytest_ldlt.log 2>&1
echo "pytest exit: $?"; tail -5 gpurun_out/pytest_ldlt.log
for sp in 7 0 1 3 5; do
  echo "=== HIOPAMD_DF_SPINE=$sp ==="
  HIOPAMD_DF_SPINE=$sp DF_TIMELINE=0 timeout -s KILL 120 python scripts/df_stamps.py 2>&1 | tail -2
done
echo "=== stepwise path, F16 mfma vs valu ==="
HIOPAMD_DF=0 DF_TIMELINE=0 timeout -s KILL 120 python scripts/df_stamps.py 2>&1 | tail -1
HIOPAMD_DF=0 HIOPAMD_F16=0 DF_TIMELINE=0 timeout -s KILL 120 python scripts/df_stamps.py 2>&1 | tail -1
echo "=== stamps (SPINE=7) ==="
HIOPAMD_DF_SPINE=7 DF_MODES=1 timeout -s KILL 120 python scripts/df_stamps.py > gpurun_out/stamps7.log 2>&1; tail -40 gpurun_out/stamps7.log
