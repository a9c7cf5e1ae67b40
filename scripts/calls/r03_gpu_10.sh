#!/bin/bash
# round 3, GPU call 10: one V workspace per super-panel; near/far queues and K = 512 on top — timing matrix + timeline
set -u
mkdir -p gpurun_out/r03_10
export TMPDIR=/tmp
O=gpurun_out/r03_10
timeout 900 python -m pytest tests/test_gpu_ldlt_kkt.py -m gpu -q -x -k "dataflow or full_size or ragged or equals" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest.log
for cfg in "HIOPAMD_DF_SPLIT=1 HIOPAMD_DF_K512=1073741824" "HIOPAMD_DF_SPLIT=1 HIOPAMD_DF_K512=0" "HIOPAMD_DF_SPLIT=0 HIOPAMD_DF_K512=0" "HIOPAMD_DF_SPLIT=0 HIOPAMD_DF_K512=0 HIOPAMD_DF_NVB=4" "HIOPAMD_DF_SPLIT=1 HIOPAMD_DF_K512=16" "HIOPAMD_DF_SPLIT=0 HIOPAMD_DF_K512=16" "HIOPAMD_DF_SPLIT=1 HIOPAMD_DF_K512=10"; do
  echo "=== $cfg"
  env $cfg DF_TIMELINE=0 timeout -s KILL 180 python scripts/df_stamps.py 2>&1 | tail -1
done
HIOPAMD_DF_SPLIT=1 DF_TIMELINE=1 DF_MODES=1 timeout -s KILL 180 python scripts/df_stamps.py > $O/timeline_default.txt 2>&1
head -40 $O/timeline_default.txt | tail -37
