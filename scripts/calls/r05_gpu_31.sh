#!/bin/bash
# Round 5, call 29: the pivoted solve's sweeps as a HIP graph: pivoted-mode tests + timing (first solve eager, second captures, later replay)
set -u
export TMPDIR=/tmp
O=gpurun_out/r05_31
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "bk or pivot or BuKa or buka" 2>&1 | tail -3 | tee $O/pytest.log
timeout 300 python scripts/bk_time.py 2048 8192 2>&1 | grep -v amdgpu.ids | tee $O/bk_time.txt
