#!/bin/bash
# round 3, GPU call 49: soak until a time-out; where in the tile loop are the workgroups that the snapshot shows "in the tile loop"?
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03_49
env DF_TIMELINE=0 timeout -s KILL 180 python scripts/df_stamps.py 2>&1 | tail -1
for i in $(seq 1 30); do
env HIOPAMD_DF_DEBUG=1 DF_REPS=400 DF_OBJECTS=4 timeout -s KILL 300 python scripts/df_repeat.py > gpurun_out/r03_49/soak_$i.log 2>&1; rc=$?
echo "soak $i exit $rc: $(tail -1 gpurun_out/r03_49/soak_$i.log | cut -c1-160)"
if grep -q "timed out" gpurun_out/r03_49/soak_$i.log; then grep "in the tile loop at stage\|bounded wait\|wide kernel:" gpurun_out/r03_49/soak_$i.log | cut -c1-200 | head -50; break; fi
done
