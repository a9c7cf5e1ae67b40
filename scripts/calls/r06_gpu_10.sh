#!/bin/bash
# Round 6, call 10: threads per front of the sparse factor kernel (128 threads up to 48 / 64 / 96 rows)
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_10
mkdir -p $O
cp hiop_amd/lib/libhiopamd.so /tmp/shipped.so
for v in tri tri_t64 tri_t96 tri tri_t64 tri_t96; do
  cp build_variants/$v.so hiop_amd/lib/libhiopamd.so
  echo "== $v" | tee -a $O/sparse_time.txt
  python scripts/sparse_ldl_time.py 1000000 5 2>&1 | tail -1 | tee -a $O/sparse_time.txt
  python scripts/sparse_ldl_time.py 200000 20 2>&1 | tail -1 | tee -a $O/sparse_time.txt
done
cp /tmp/shipped.so hiop_amd/lib/libhiopamd.so
