#!/bin/bash
# Round 5, call 24: where a column step of the pivoted panel kernel spends its time (instrumented build, workgroup 0's clock)
set -u
export TMPDIR=/tmp
O=gpurun_out/r05_24
mkdir -p $O
cp hiop_amd/lib/libhiopamd.so /tmp/shipped.so
cp build_variants/bktime.so hiop_amd/lib/libhiopamd.so
timeout 300 python scripts/bk_time.py 8192 2>&1 | grep -v amdgpu.ids | tee $O/bk_phases.txt
cp /tmp/shipped.so hiop_amd/lib/libhiopamd.so
