#!/bin/bash
# Round 5, call 41: sparse LDL^T with column-major L panels (coalesced sweeps) and the division-free rank-1 update: parity, per-level times, bench entries
set -u
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_sparse_ldl.py tests/test_gpu_kkt_sparse.py -x -q 2>&1 | grep "passed\|failed\|rror" | tail -3
bash scripts/calls/r05_gpu_40.sh 2>&1 | grep "sl_\|exit"
bash scripts/calls/r05_gpu_33b.sh 2>&1 | tail -2
