#!/bin/bash
# Round 6, call 29: the added safe-mode case at a padded order; then a long soak of the shipped shape at N = 8192 on the final library
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_29
mkdir -p $O
sha256sum hiop_amd/lib/libhiopamd.so | tee $O/summary.txt
timeout 600 python -m pytest tests/test_gpu_ldlt_kkt.py -m gpu -x -q -p no:cacheprovider -k "safe_mode or block_boundaries" 2>&1 | grep -E "passed|failed" | tee -a $O/summary.txt
for i in 1 2 3 4 5 6 7 8; do
  env DF_RETRY_COPY=1 DF_REPS=2500 DF_OBJECTS=4 timeout -s KILL 600 python scripts/df_repeat.py > $O/chunk_$i.log 2>&1; rc=$?
  echo "N = 8192: exit $rc: $(tail -1 $O/chunk_$i.log | cut -c1-200)" | tee -a $O/summary.txt
  [ $rc -ne 0 ] && break
done
exit 0
