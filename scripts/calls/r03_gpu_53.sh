#!/bin/bash
# round 3, GPU call 53: the soak with ONE workgroup of the wide kernel per CU (HIOPAMD_DF_WGS=240: every CU keeps > 80 KB of LDS free) —
# if the freezes come from a restore that cannot put two 73.7 KB workgroups back on a CU, they should be gone
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03_53
for i in $(seq 1 26); do
env HIOPAMD_DF_WGS=240 HIOPAMD_DF_DEBUG=1 DF_REPS=400 DF_OBJECTS=4 timeout -s KILL 300 python scripts/df_repeat.py > gpurun_out/r03_53/soak_$i.log 2>&1; rc=$?
echo "soak $i exit $rc: $(tail -1 gpurun_out/r03_53/soak_$i.log | cut -c1-140)"
if grep -q "timed out" gpurun_out/r03_53/soak_$i.log; then grep "bounded wait\|failed after\|in the tile loop at stage\|wide kernel:" gpurun_out/r03_53/soak_$i.log | cut -c1-200 | head -12; break; fi
done
