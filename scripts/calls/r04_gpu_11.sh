#!/bin/bash
# Round 4, call 11: A/B of the load forms of y = J x (HIOPAMD_GEMV_OPT bits: 1 plain loads, 2 row batches of 4, 4 two chunks per block,
# 32 guarded form), x = J^T y (8: non-temporal) and the secant pass (16: non-temporal) inside one call
set -u
O=gpurun_out/r04_11; mkdir -p $O
for v in 0 1 2 4 6 8 16 24 32 0; do
  echo "--- HIOPAMD_GEMV_OPT=$v"; HIOPAMD_GEMV_OPT=$v STEPS=15 timeout -s KILL 300 python scripts/dense_step_time.py 2>&1 | grep -v amdgpu.ids | tee -a $O/dense_opt$v.log
done
