#!/bin/bash
# Round 5, call 37: soak of the pivoted panel kernel's protocol (scripts/bk_repeat.py)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r05_37
timeout -s KILL 500 python scripts/bk_repeat.py 1100 1500 2048 1500 8192 400 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_37/bk_soak.txt
