#!/bin/bash
# per-task time stamps of the dataflow solve (HIOPAMD_SOLVE_STAMPS), N = 8192
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
HIOPAMD_SOLVE_STAMPS=1 DF_TIMELINE=0 timeout -s KILL 120 python scripts/df_stamps.py > gpurun_out/solve_stamps.log 2>&1
grep "hiop_amd\] solve" gpurun_out/solve_stamps.log | tail -6; tail -2 gpurun_out/solve_stamps.log
