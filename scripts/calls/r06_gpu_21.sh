#!/bin/bash
# Round 6, call 21: odd orders padded, the padded copy's lower triangle defined: the poison-build gate groups + LDL^T / sparse / KKT tests
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_21
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_poisoned_allocations.py tests/test_gpu_sparse_ldl.py tests/test_gpu_kkt_sparse.py tests/test_gpu_lowrank.py tests/test_gpu_kkt_xycyd.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -15 | tee $O/pytest.txt
HIOPAMD_LDLT_PAD=0 timeout 300 python scripts/factor_time.py 8191 2>&1 | tail -2 | tee -a $O/factor_time.txt
timeout 300 python scripts/factor_time.py 8191 2>&1 | tail -2 | tee -a $O/factor_time.txt
exit 0
