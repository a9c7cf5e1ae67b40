#!/bin/bash
# round 3, GPU call 47: forced time-out with the debug dump (is the waiter's snapshot of the state words populated?), LDL^T tests
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03_47
HIOPAMD_DF_TIMEOUT_MS=0.001 HIOPAMD_DF_DEBUG=1 DF_N=4096 DF_REPS=1 DF_OBJECTS=1 timeout 120 python scripts/df_repeat.py 2>&1 | grep "hiop_amd" | grep -v "chain role [0-9]\|writers of\|holds subst" | cut -c1-200 | head -14
timeout 600 python -m pytest tests/test_gpu_ldlt_timeout_recovery.py tests/test_gpu_ldlt_kkt.py -m gpu -q -x > gpurun_out/r03_47/pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r03_47/pytest.log
