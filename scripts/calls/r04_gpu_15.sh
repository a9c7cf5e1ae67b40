#!/bin/bash
# Round 4, call 15: the pivoted (Bunch-Kaufman) factorisation: parity tests, then time at n = 2048 / 8192
set -u
O=gpurun_out/r04_15; mkdir -p $O
timeout -s KILL 900 python -m pytest tests/test_gpu_ldlt_bk.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest exit $?: $(grep -E 'passed|failed' $O/pytest.log | tail -1)"
grep -E "^(FAILED|ERROR)|Error|assert|^E " $O/pytest.log | head -40
timeout -s KILL 300 python scripts/bk_time.py 2048 8192 2>&1 | grep -v amdgpu.ids | tee $O/bk_time.log
