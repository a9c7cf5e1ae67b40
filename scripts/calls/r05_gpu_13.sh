#!/bin/bash
# round 5, call 13: the pivoted (Bunch-Kaufman) factorisation with ONE launch per 64-column panel (grid barriers between the phases)
set -u
mkdir -p gpurun_out/r05_13
export TMPDIR=/tmp
echo "=== pytest bk ==="
timeout 900 python -m pytest tests/test_gpu_ldlt_bk.py tests/test_gpu_kkt_xycyd.py -q -x > gpurun_out/r05_13/pytest.log 2>&1; echo "pytest exit: $?"
tail -8 gpurun_out/r05_13/pytest.log
echo "=== time ==="
timeout 300 python scripts/bk_time.py 2048 8192 > gpurun_out/r05_13/bk_time.txt 2>&1; cat gpurun_out/r05_13/bk_time.txt | grep -v amdgpu.ids
