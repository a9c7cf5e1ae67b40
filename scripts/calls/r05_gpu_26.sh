#!/bin/bash
# Round 5, call 26: pivoted LDL^T with the interchanges of the columns in front of the panel applied once per panel: parity, timing, phases
set -u
export TMPDIR=/tmp
O=gpurun_out/r05_26
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ldlt_bk.py -x -q 2>&1 | tail -3 | tee $O/pytest.log
timeout 300 python scripts/bk_time.py 2048 8192 2>&1 | grep -v amdgpu.ids | tee $O/bk_time.txt
cp hiop_amd/lib/libhiopamd.so /tmp/shipped.so
cp build_variants/bktime.so hiop_amd/lib/libhiopamd.so
timeout 300 python scripts/bk_time.py 8192 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/bk_phases.txt
cp /tmp/shipped.so hiop_amd/lib/libhiopamd.so
