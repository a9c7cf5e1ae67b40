#!/bin/bash
# Round 4, call 22: the 8-byte forms of y = A x and x = A^T y (odd leading dimension) with unguarded / batched loads: parity, then
# the end-to-end IPM solve on the stock MdsEx1 (n_dense = 4097: odd) twice
set -u
O=gpurun_out/r04_22; mkdir -p $O
timeout -s KILL 900 python -m pytest tests/test_gpu_dense_sparse.py tests/test_reference_known_answers.py tests/test_gpu_ldlt_kkt.py tests/test_gpu_ipm_device.py tests/test_c_interface.py tests/test_gpu_ldlt_bk.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest exit $?: $(grep -E 'passed|failed' $O/pytest.log | tail -1)"
grep -E "^(FAILED|ERROR)|Error|assert|^E " $O/pytest.log | head -20
for i in 1 2; do timeout 300 python -c "
import bench
d = bench.ipm_end_to_end_bench()
print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in d['device'].items()}, flush=True)" 2>&1 | grep objective | tee -a $O/e2e.log; done
