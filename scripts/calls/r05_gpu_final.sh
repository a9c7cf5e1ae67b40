#!/bin/bash
# Round 5, final pass on the shipped library: whole GPU suite, smoke, bench + rocprofv3 kernel statistics of the same command, dense-kernel
# PMC passes again (y = J x changed since the first pass), pivoted-mode timing.  (The LDL^T PMC passes of r05_gpu_full.sh stay valid: the
# dataflow kernels did not change.)
set -u
export TMPDIR=/tmp
SKIP_PMC=1 bash scripts/calls/r05_gpu_full.sh
bash scripts/calls/r05_pmc.sh dense_only 2>&1 | grep "hbm_bytes_per_launch\|exit"
timeout -s KILL 300 python scripts/bk_time.py 2048 8192 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_full/bk_time.log
