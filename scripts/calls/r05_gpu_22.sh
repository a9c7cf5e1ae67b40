#!/bin/bash
# Round 5, call 22: pivoted panel kernel, XCD-local form (participants chosen at run time on one XCD; L2-kept stores) against the agent-scope form
set -u
export TMPDIR=/tmp
O=gpurun_out/r05_22
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ldlt_bk.py -x -q 2>&1 | tail -3 | tee $O/pytest.log
for m in 1 0 1 0; do
  echo "HIOPAMD_BK_LOCAL=$m"
  HIOPAMD_BK_LOCAL=$m timeout 300 python scripts/bk_time.py 2048 8192 2>&1 | grep -v amdgpu.ids | tee -a $O/bk_time_local$m.txt
done
