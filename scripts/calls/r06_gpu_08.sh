#!/bin/bash
# Round 6, call 8: sparse LDL^T: backward sweep in axpy form (shipped) and leaf-supernode widths 24 / 32 against 48
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_08
mkdir -p $O
cp hiop_amd/lib/libhiopamd.so /tmp/shipped.so
for v in cur leaf32 leaf24; do
  cp build_variants/$v.so hiop_amd/lib/libhiopamd.so
  echo "== $v" | tee -a $O/sparse_time.txt
  python scripts/sparse_ldl_time.py 1000000 5 2>&1 | tail -1 | tee -a $O/sparse_time.txt
  python scripts/sparse_ldl_time.py 1000000 1 2>&1 | tail -1 | tee -a $O/sparse_time.txt
  python scripts/sparse_ldl_time.py 200000 20 2>&1 | tail -1 | tee -a $O/sparse_time.txt
done
cp /tmp/shipped.so hiop_amd/lib/libhiopamd.so
