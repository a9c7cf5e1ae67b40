#!/bin/bash
# Round 4, call 16: timesMat / transTimesMat / timesMatTrans on fp64 MFMA tiles: parity of the GEMM family and of the paths that use it, timing
set -u
O=gpurun_out/r04_16; mkdir -p $O
timeout -s KILL 900 python -m pytest tests/test_gpu_dense_sparse.py tests/test_gpu_lowrank.py tests/test_reference_known_answers.py tests/test_gpu_full_size.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest exit $?: $(grep -E 'passed|failed' $O/pytest.log | tail -1)"
grep -E "^(FAILED|ERROR)|Error|assert|^E " $O/pytest.log | head -30
timeout 120 python scripts/gemm_time.py 2>&1 | grep -v amdgpu.ids | tee $O/gemm.log
HIOPAMD_GEMM=0 timeout 120 python scripts/gemm_time.py 2>&1 | grep -v amdgpu.ids | tee -a $O/gemm.log
