#!/bin/bash
# Round 4, call 13: dense low-rank step after the 2l x 2l LU is factored once per secant update, the four l x l Gram blocks are one
# pass, fewer small copies; A/B of one / two column pairs per thread in x = J^T y and in the secant pass (HIOPAMD_CP)
set -u
O=gpurun_out/r04_13; mkdir -p $O
timeout -s KILL 900 python -m pytest tests/test_gpu_lowrank.py tests/test_gpu_full_size.py tests/test_gpu_two_rank.py tests/test_gpu_dense_sparse.py tests/test_gpu_ipm_device.py tests/test_c_interface.py tests/test_reference_known_answers.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest exit $?: $(grep -E 'passed|failed' $O/pytest.log | tail -1)"
grep -E "^(FAILED|ERROR)|Error|assert" $O/pytest.log | head -20
for v in 1 2 1 2; do
  echo "--- HIOPAMD_CP=$v"; HIOPAMD_CP=$v STEPS=15 timeout -s KILL 300 python scripts/dense_step_time.py 2>&1 | grep -v amdgpu.ids | tee -a $O/dense_cp$v.log
done
