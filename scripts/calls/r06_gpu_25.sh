#!/bin/bash
# Round 6, call 25: leaf width of the nested dissection (48 / 32 / 28 columns) now that small leaves are factored in registers
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_25
mkdir -p $O
cp hiop_amd/lib/libhiopamd.so /tmp/shipped.so
for v in leaf48 leaf32 leaf28 leaf48 leaf32 leaf28; do
  cp build_variants/$v.so hiop_amd/lib/libhiopamd.so
  echo "== $v" | tee -a $O/sparse_time.txt
  for pat in "1000000 5" "200000 5" "700000 5" "1000000 3" "1000000 7" "500000 10" "200000 20"; do
    timeout 300 python scripts/sparse_ldl_time.py $pat 2>&1 | tail -1 | tee -a $O/sparse_time.txt
  done
done
cp /tmp/shipped.so hiop_amd/lib/libhiopamd.so
exit 0
