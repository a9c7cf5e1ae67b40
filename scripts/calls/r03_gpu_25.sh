#!/bin/bash
# round 3, GPU call 25: second workgroups of every CU leave the wide kernel at super-panel HIOPAMD_DF_RETIRE
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03_25
for r in 99 16 12 14 18 20 99 16; do
echo "=== HIOPAMD_DF_RETIRE=$r"
env HIOPAMD_DF_RETIRE=$r DF_TIMELINE=0 timeout -s KILL 180 python scripts/df_stamps.py 2>&1 | tail -1
done
timeout 900 python -m pytest tests/test_gpu_ldlt_kkt.py -m gpu -q -x > gpurun_out/r03_25/pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r03_25/pytest.log
