#!/bin/bash
# Round 4, call 18: y = A x with the fold in the last workgroup to arrive (one launch) against two launches (HIOPAMD_GEMV_FOLD=2);
# non-temporal stores of J_prev in the secant pass (HIOPAMD_SJ_NT=1); parity of everything that uses the GEMV
set -u
O=gpurun_out/r04_18; mkdir -p $O
timeout -s KILL 900 python -m pytest tests/test_gpu_dense_sparse.py tests/test_gpu_lowrank.py tests/test_gpu_full_size.py tests/test_gpu_two_rank.py tests/test_gpu_ldlt_bk.py tests/test_gpu_kkt_xycyd.py tests/test_reference_known_answers.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest exit $?: $(grep -E 'passed|failed' $O/pytest.log | tail -1)"
grep -E "^(FAILED|ERROR)|Error|assert|^E " $O/pytest.log | head -20
for v in "X=0" "HIOPAMD_GEMV_FOLD=2" "X=0" "HIOPAMD_GEMV_FOLD=2"; do
  echo "--- $v"; env $v STEPS=15 timeout -s KILL 300 python scripts/dense_step_time.py 2>&1 | grep -v amdgpu.ids | tee -a $O/dense.log
done
