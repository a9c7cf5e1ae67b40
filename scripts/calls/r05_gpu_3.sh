#!/bin/bash
# round 5, call 3: the general sparse LDL^T (f2) on hardware + the micro-optimised 16 x 16 factor
set -u
mkdir -p gpurun_out/r05_3
export TMPDIR=/tmp
echo "=== pytest sparse ==="
timeout 1200 python -m pytest tests/test_gpu_sparse_ldl.py tests/test_gpu_kkt_sparse.py tests/test_gpu_csr_condensed.py -q -x --durations=8 > gpurun_out/r05_3/pytest.log 2>&1; echo "pytest exit: $?"
tail -30 gpurun_out/r05_3/pytest.log
echo "=== stamps ==="
timeout 200 python scripts/df_stamps.py > gpurun_out/r05_3/stamps.log 2>&1; echo "stamps exit: $?"
grep -v "^  [ 0-9][0-9] |" gpurun_out/r05_3/stamps.log | grep "spine steps\|inside F\|matrixChanged\|F(p) published"
