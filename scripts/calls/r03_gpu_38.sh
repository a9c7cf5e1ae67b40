#!/bin/bash
# round 3, GPU call 38: one V workspace per super-panel + stronger polling back-off: tests, timing, long soak
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03_38
timeout 600 python -m pytest tests/test_gpu_ldlt_kkt.py tests/test_gpu_ldlt_timeout_recovery.py -m gpu -q -x > gpurun_out/r03_38/pytest.log 2>&1; echo "pytest exit $?"; tail -2 gpurun_out/r03_38/pytest.log
for cfg in "HIOPAMD_DF_NVB=4" "HIOPAMD_DF_NVB=99" "HIOPAMD_DF_NVB=4" "HIOPAMD_DF_NVB=99"; do echo "== $cfg"; env $cfg DF_TIMELINE=0 timeout -s KILL 180 python scripts/df_stamps.py 2>&1 | tail -1; done
for i in $(seq 1 36); do
env HIOPAMD_DF_DEBUG=1 DF_REPS=400 DF_OBJECTS=4 timeout -s KILL 300 python scripts/df_repeat.py > gpurun_out/r03_38/soak_$i.log 2>&1; rc=$?
echo "soak $i exit $rc: $(tail -1 gpurun_out/r03_38/soak_$i.log | cut -c1-160)"
if grep -q "timed out" gpurun_out/r03_38/soak_$i.log; then grep "hiop_amd\|failed after" gpurun_out/r03_38/soak_$i.log | grep -v "chain role [0-9]\|holds subst" | cut -c1-260 | head -40; break; fi
done
