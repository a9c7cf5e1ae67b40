#!/bin/bash
# Round 6, call 22: the sparse tests (normal + poison build) after the test's patterns were chosen inside the register kernel's range
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_22
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sparse_ldl.py -x -q -p no:cacheprovider 2>&1 | tail -6 | tee $O/pytest.txt
timeout 1500 python -m pytest tests/test_gpu_poisoned_allocations.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -6 | tee -a $O/pytest.txt
exit 0
