#!/bin/bash
# round 3, GPU call 48: the same soak with a 20 s wait limit — does the stall resolve by itself (a slow phase of the device), and how long is it?
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03_48
for i in $(seq 1 34); do
env HIOPAMD_DF_TIMEOUT_MS=20000 DF_REPS=400 DF_OBJECTS=4 timeout -s KILL 400 python scripts/df_repeat.py > gpurun_out/r03_48/soak_$i.log 2>&1; rc=$?
echo "soak $i exit $rc: $(tail -1 gpurun_out/r03_48/soak_$i.log | cut -c1-200)"
grep "took\|timed out\|failed after" gpurun_out/r03_48/soak_$i.log | cut -c1-200 | head -5
if grep -q "took\|timed out" gpurun_out/r03_48/soak_$i.log; then break; fi
done
