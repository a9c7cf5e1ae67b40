#!/bin/bash
# Round 5, call 28: pivoted LDL^T, shipped shape 16 x 512 with the fused remainder loop: all pivoted-mode tests (solver, KKT back-end, adapters' class) + timing
set -u
export TMPDIR=/tmp
O=gpurun_out/r05_28
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "bk or pivot or BuKa or buka" 2>&1 | tail -3 | tee $O/pytest.log
timeout 300 python scripts/bk_time.py 2048 8192 2>&1 | grep -v amdgpu.ids | tee $O/bk_time.txt
