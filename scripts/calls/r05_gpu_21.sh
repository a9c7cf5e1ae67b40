#!/bin/bash
# Round 5, call 21: the pivoted panel kernel with published decision scalars (one barrier less per two-phase column step, decisions
# fetched in one round trip by the deciding wave, 16 loads in flight, phase C loads hoisted): parity tests + timing
set -u
export TMPDIR=/tmp
O=gpurun_out/r05_21
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ldlt_bk.py -x -q 2>&1 | tail -3 | tee $O/pytest.log
timeout 300 python scripts/bk_time.py 2048 8192 2>&1 | grep -v amdgpu.ids | tee $O/bk_time.txt
