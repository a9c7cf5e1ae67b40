#!/bin/bash
# Round 6, call 13: host-side symbolic analysis on several threads (nested dissection forks at its top levels, contribution sorts per front in parallel)
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_13
mkdir -p $O
nproc | tee $O/host.txt
cp hiop_amd/lib/libhiopamd.so /tmp/shipped.so
for v in par par2 par par2; do
  cp build_variants/$v.so hiop_amd/lib/libhiopamd.so
  echo "== $v" | tee -a $O/sparse_time.txt
  HIOPAMD_SL_TIMING=1 python scripts/sparse_ldl_time.py 1000000 5 2>&1 | grep -E "sparse analysis|^n " | tee -a $O/sparse_time.txt
done
cp /tmp/shipped.so hiop_amd/lib/libhiopamd.so
timeout 900 python -m pytest tests/test_gpu_sparse_ldl.py tests/test_gpu_kkt_sparse.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
