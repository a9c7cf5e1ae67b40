#!/bin/bash
# Round 6, call 15: sparse LDL^T — groups of levels in one launch + fronts in registers: tests, then timing against the per-level form
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_15
mkdir -p $O
sha256sum hiop_amd/lib/libhiopamd.so | tee $O/library.txt
timeout 900 python -m pytest tests/test_gpu_sparse_ldl.py -x -q -p no:cacheprovider 2>&1 | tail -5 | tee $O/pytest_sparse.txt
HIOPAMD_SL_REGS=0 timeout 900 python -m pytest tests/test_gpu_sparse_ldl.py -x -q -p no:cacheprovider 2>&1 | tail -3 | tee -a $O/pytest_sparse.txt
HIOPAMD_BUILD_VARIANT=poison timeout 900 python -m pytest tests/test_gpu_sparse_ldl.py -x -q -p no:cacheprovider 2>&1 | tail -3 | tee -a $O/pytest_sparse.txt
for cfg in "1 0" "1 1" "6 0" "6 1" "4 1" "8 1" "1 0" "6 1"; do
  set -- $cfg
  echo "== depth $1 regs $2" | tee -a $O/sparse_time.txt
  HIOPAMD_SL_GROUP_DEPTH=$1 HIOPAMD_SL_REGS=$2 timeout 300 python scripts/sparse_ldl_time.py 1000000 5 2>&1 | tail -1 | tee -a $O/sparse_time.txt
  HIOPAMD_SL_GROUP_DEPTH=$1 HIOPAMD_SL_REGS=$2 timeout 300 python scripts/sparse_ldl_time.py 200000 20 2>&1 | tail -1 | tee -a $O/sparse_time.txt
done
exit 0
