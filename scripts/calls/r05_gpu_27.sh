#!/bin/bash
# Round 5, call 27: pivoted panel kernel as 16 workgroups of 512 threads (32 or 16 row entries in flight) against 8 x 1024 (16 in flight)
set -u
export TMPDIR=/tmp
O=gpurun_out/r05_27
mkdir -p $O
cp hiop_amd/lib/libhiopamd.so /tmp/shipped.so
for v in g8 g16 g16d16 g8 g16; do
  cp build_variants/$v.so hiop_amd/lib/libhiopamd.so
  echo "== $v"
  timeout 300 python scripts/bk_time.py 2048 8192 2>&1 | grep -v amdgpu.ids | tee -a $O/bk_time_$v.txt
done
cp build_variants/g16.so hiop_amd/lib/libhiopamd.so
timeout 600 python -m pytest tests/test_gpu_ldlt_bk.py -x -q 2>&1 | tail -2 | tee $O/pytest_g16.log
cp /tmp/shipped.so hiop_amd/lib/libhiopamd.so
