#!/bin/bash
# Round 5, call 23: pivoted panel kernel with tagged granules instead of the barriers behind phases A and B; XCD-local against agent-scope form
set -u
export TMPDIR=/tmp
O=gpurun_out/r05_23
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ldlt_bk.py -x -q 2>&1 | tail -3 | tee $O/pytest.log
for m in 1 0 1 0; do
  echo "HIOPAMD_BK_LOCAL=$m"
  HIOPAMD_BK_LOCAL=$m timeout 300 python scripts/bk_time.py 2048 8192 2>&1 | grep -v amdgpu.ids | tee -a $O/bk_time_local$m.txt
done
