#!/bin/bash
# Round 5, call 35: the sparse LDL^T tests incl. the sparse solver adapter's C-ABI recipe
set -u
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_sparse_ldl.py -x -q 2>&1 | grep "passed\|failed\|rror" | tail -3
