#!/bin/bash
# Round 6, call 9: sparse LDL^T fronts as packed lower triangles in LDS (half the LDS per front): tests, then timing against the square form
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_09
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sparse_ldl.py tests/test_gpu_kkt_sparse.py -m gpu -q -x -p no:cacheprovider > $O/pytest_sparse.log 2>&1
echo "pytest(sparse) exit: $?"; tail -4 $O/pytest_sparse.log
cp hiop_amd/lib/libhiopamd.so /tmp/shipped.so
for v in cur tri cur tri; do
  cp build_variants/$v.so hiop_amd/lib/libhiopamd.so
  echo "== $v" | tee -a $O/sparse_time.txt
  python scripts/sparse_ldl_time.py 1000000 5 2>&1 | tail -1 | tee -a $O/sparse_time.txt
  python scripts/sparse_ldl_time.py 200000 20 2>&1 | tail -1 | tee -a $O/sparse_time.txt
done
cp /tmp/shipped.so hiop_amd/lib/libhiopamd.so
