#!/bin/bash
# round 3, GPU call 30: recovery from expired waits (forced), the LDL^T tests, timing with the retiring workgroups
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03_30
timeout 900 python -m pytest tests/test_gpu_ldlt_timeout_recovery.py tests/test_gpu_ldlt_kkt.py -m gpu -q -x > gpurun_out/r03_30/pytest.log 2>&1; echo "pytest exit $?"; tail -5 gpurun_out/r03_30/pytest.log
for cfg in "HIOPAMD_DF_RETIRE=99" "HIOPAMD_DF_RETIRE=16" "HIOPAMD_DF_RETIRE=14" "HIOPAMD_DF_RETIRE=99" "HIOPAMD_DF_RETIRE=16"; do
echo "=== $cfg"; env $cfg DF_TIMELINE=0 timeout -s KILL 180 python scripts/df_stamps.py 2>&1 | tail -1
done
for i in 1 2 3; do
env DF_REPS=400 DF_OBJECTS=4 timeout -s KILL 300 python scripts/df_repeat.py > gpurun_out/r03_30/soak_$i.log 2>&1; rc=$?
echo "soak $i exit $rc: $(tail -1 gpurun_out/r03_30/soak_$i.log | cut -c1-160)"; grep "hiop_amd\|failed after" gpurun_out/r03_30/soak_$i.log | cut -c1-250 | head -5
done
