#!/bin/bash
# Round 6, call 3: the whole gate (GPU suite incl. the poison-build runs, smoke, bench + kernel trace) on the fixed library
set -u
export TMPDIR=/tmp
bash scripts/gpu_check.sh
mkdir -p gpurun_out/r06_03 && cp gpurun_out/pytest_gpu.log gpurun_out/bench.json gpurun_out/smoke.log gpurun_out/r06_03/ 2>/dev/null
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r06_03/bench_kernel_stats.csv
