#!/bin/bash
# round 3, GPU call 7: K = 512 fused update tasks (DF_UP2) — parity of the factorisation, timing on / off
set -u
mkdir -p gpurun_out/r03_7
export TMPDIR=/tmp
O=gpurun_out/r03_7
timeout 1200 python -m pytest tests/test_gpu_ldlt_kkt.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -6 $O/pytest.log
for k in 1073741824 0 16 20 24; do
  echo "=== HIOPAMD_DF_K512=$k"
  HIOPAMD_DF_K512=$k DF_TIMELINE=0 timeout -s KILL 180 python scripts/df_stamps.py 2>&1 | tail -1
done
