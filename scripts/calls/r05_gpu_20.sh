#!/bin/bash
# Round 5, call 20: A/B of the substitution task with all operand loads of a block row in one batch (hoist) against the shipped library
# (base), phase sums of both; the dense-kernel PMC passes again (the first attempt imported the wrong module).
set -u
export TMPDIR=/tmp
O=gpurun_out/r05_20
mkdir -p $O
cp hiop_amd/lib/libhiopamd.so /tmp/shipped.so
STEPS=20 bash scripts/ab_bench.sh base hoist 2>&1 | tee $O/ab.txt
for v in base hoist; do
  cp build_variants/$v.so hiop_amd/lib/libhiopamd.so
  echo "== stamps $v"
  HIOPAMD_DF_STAMPS=1 timeout 300 python scripts/df_stamps.py 2>&1 | grep "wide kernel phases\|spine steps\|matrixChanged\|residual" | tee -a $O/stamps_$v.txt
  timeout 300 python -m pytest tests/test_gpu_ldlt_kkt.py -x -q 2>&1 | tail -1
done
cp /tmp/shipped.so hiop_amd/lib/libhiopamd.so
bash scripts/calls/r05_pmc.sh dense_only 2>&1 | tail -40
