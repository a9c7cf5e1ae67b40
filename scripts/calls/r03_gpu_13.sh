#!/bin/bash
# A/B on one box: Gram stage-buffer layout (planes vs padded rows), GEMV stage 1 (round 3 vs round 2)
set -u
export TMPDIR=/tmp
for rep in 1 2; do
for cfg in "HIOPAMD_GRAM_LDS=1 HIOPAMD_GEMV=0" "HIOPAMD_GRAM_LDS=0 HIOPAMD_GEMV=1"; do
  echo "=== $cfg"; env $cfg timeout 300 python scripts/calls/r03_ab_dense.py 2>&1 | tail -1
done
done
