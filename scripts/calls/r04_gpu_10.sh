#!/bin/bash
# Round 4, call 10: the dense low-rank step after (a) solveCompressed with ONE collective (the low-rank inverse multiplied out against J),
# (b) the secant update's scalars in one host round trip, (c) unguarded interior loads in y = J x, batched loads in the secant
# pass and in the Gram fold.  Parity tests of the touched paths, then A/B of the GEMV forms inside one call.
set -u
O=gpurun_out/r04_10; mkdir -p $O
timeout -s KILL 900 python -m pytest tests/test_gpu_lowrank.py tests/test_gpu_full_size.py tests/test_gpu_two_rank.py tests/test_gpu_dense_sparse.py tests/test_gpu_ipm_device.py tests/test_c_interface.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest exit $?: $(tail -1 $O/pytest.log)"
grep -E "^(FAILED|ERROR)|Error|assert" $O/pytest.log | head -20
for v in 0 3 2 0; do
  echo "--- HIOPAMD_GEMV=$v"; HIOPAMD_GEMV=$v timeout -s KILL 300 python scripts/dense_step_time.py 2>&1 | grep -v amdgpu.ids | tee -a $O/dense_gemv$v.log
done
