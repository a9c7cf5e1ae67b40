#!/bin/bash
# Round 5, call 25: substitution task with the next block column's operands requested ahead (two buffers) against the shipped library
set -u
export TMPDIR=/tmp
O=gpurun_out/r05_25
mkdir -p $O
cp hiop_amd/lib/libhiopamd.so /tmp/shipped.so
STEPS=20 bash scripts/ab_bench.sh base trpf 2>&1 | tee $O/ab.txt
for v in base trpf; do
  cp build_variants/$v.so hiop_amd/lib/libhiopamd.so
  echo "== stamps $v"
  HIOPAMD_DF_STAMPS=1 timeout 300 python scripts/df_stamps.py 2>&1 | grep "wide kernel phases\|spine steps\|matrixChanged\|residual" | tee -a $O/stamps_$v.txt
done
cp build_variants/trpf.so hiop_amd/lib/libhiopamd.so
timeout 300 python -m pytest tests/test_gpu_ldlt_kkt.py -x -q 2>&1 | tail -1
cp /tmp/shipped.so hiop_amd/lib/libhiopamd.so
